"""GPU test of the row-slab mode with the real HIP backend.  The GPU box has one card, so the
ranks are threads of one process talking through an in-process stand-in for torch.distributed's
P2P calls (same call sequence as the RCCL path: batch_isend_irecv of P2POp(isend / irecv))."""
import queue
import threading

import numpy as np
import pytest

from opticalflowhs_amd import slab, synth

pytestmark = pytest.mark.gpu


class LocalDist(object):
    """Minimal in-process torch.distributed look-alike for one rank (threads as ranks)."""
    boxes = {}
    lock = threading.Lock()

    def __init__(self, rank):
        self.rank = rank

    @classmethod
    def box(cls, src, dst):
        with cls.lock:
            return cls.boxes.setdefault((src, dst), queue.Queue())

    def isend(self, t, peer):
        return ("send", t, peer)

    def irecv(self, t, peer):
        return ("recv", t, peer)

    def P2POp(self, op, t, peer):
        return op(t, peer)

    def batch_isend_irecv(self, ops):
        import torch
        torch.cuda.synchronize()
        for kind, t, peer in ops:
            if kind == "send":
                self.box(self.rank, peer).put(t.clone())

        class W(object):
            def __init__(s, fn):
                s.fn = fn

            def wait(s):
                s.fn()
        out = []
        for kind, t, peer in ops:
            if kind == "recv":
                out.append(W(lambda t=t, peer=peer: t.copy_(self.box(peer, self.rank).get(timeout=60))))
            else:
                out.append(W(lambda: None))
        return out


@pytest.mark.parametrize("world,halo,iters,shape,overlapped", [(2, 8, 30, (300, 512), False), (3, 12, 40, (203, 700), False),
                                                               (2, 8, 30, (300, 512), True), (3, 10, 47, (260, 700), True),
                                                               (1, 9, 40, (150, 333), True)])
def test_slab_matches_single_gpu_bit_for_bit(hs, gpu_ok, world, halo, iters, shape, overlapped):
    H, W = shape
    A, B = synth.translating_pair(W, H, seed=3)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ctx.solve(lam=0.7, max_iter=iters, term_type=hs.TERM_ITER)
        uo, vo = ctx.flow()
    LocalDist.boxes = {}
    res, errs = {}, []

    def run(rank):
        try:
            if overlapped:  # two sub-slabs per rank, each context on its own torch stream, calls only enqueued
                import torch
                s = slab.OverlappedSlabSolver(LocalDist(rank), rank, world, W, H, halo,
                                              lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, 0, torch_stream=torch.cuda.Stream(), first_row=r0))
            else:
                s = slab.SlabSolver(LocalDist(rank), rank, world, W, H, halo,
                                    lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, 0, first_row=r0))
            r0, r1 = s.local_frame_rows()
            s.set_frames(A[r0:r1], B[r0:r1])
            s.solve(0.7, iters)
            res[rank] = (s.lo, s.hi) + s.owned_flow()
            s.close()
        except Exception as e:  # surfaced in the main thread
            errs.append((rank, repr(e)))

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join(120) for t in ths]
    assert not errs, errs
    u = np.zeros_like(uo)
    v = np.zeros_like(vo)
    for rank in range(world):
        lo, hi, ur, vr = res[rank]
        u[lo:hi], v[lo:hi] = ur, vr
    assert np.array_equal(u, uo) and np.array_equal(v, vo)
