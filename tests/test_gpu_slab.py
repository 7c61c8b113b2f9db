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

    # the one collective of the ITER|EPS path: an element-wise maximum over the ranks (threads here)
    world, barrier, slots = 1, None, {}

    class ReduceOp(object):
        MAX = "max"

    def get_backend(self):
        return "local"

    def all_reduce(self, t, op=None):
        import torch
        cls = type(self)
        with cls.lock:
            cls.slots[self.rank] = t.clone()
        cls.barrier.wait(timeout=120)
        m = torch.stack([cls.slots[r] for r in range(cls.world)]).max(0).values
        cls.barrier.wait(timeout=120)   # everybody has read every slot
        t.copy_(m)

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


@pytest.mark.parametrize("world,halo", [(2, 12), (3, 8)])
def test_slab_iter_eps_over_ranks_with_the_hip_backend(hs, gpu_ok, world, halo):
    """ITER|EPS over ranks (threads), HIP contexts: witness launches + hsflow_take_verdict on the fast path, hsflow_solve_probe
    when nobody vouches; the maximum over the ranks through all_reduce.  Stopping sweep and flow of the one-context solve
    (values below 1e-30 aside: the scaled state's denormal caveat), and a warm start after it."""
    W, H = 600, 260
    flat_a = np.full((H, W), 90, np.uint8)
    flat_b = flat_a.copy()
    flat_a[50:200, 100:480] = 120
    flat_b[50:200, 100:480] = 121
    cases = [(synth.translating_pair(W, H, seed=21), 1.0, 50, float(np.float32(1e-6))), ((flat_a, flat_b), 1e-3, 300, 1e-4)]

    def same(x, y):
        return bool(np.all((x == y) | ((np.abs(x) < 1e-30) & (np.abs(y) < 1e-30))))

    for (A, B), lam, budget, eps in cases:
        with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
            ctx.set_frames(A, B)
            i1 = ctx.solve(lam=lam, max_iter=budget, term_type=3, epsilon=eps)
            f1 = ctx.flow()
            i2 = ctx.solve(lam=lam, max_iter=20, term_type=3, epsilon=eps, use_previous=True)
            f2 = ctx.flow()
        LocalDist.boxes, LocalDist.slots, LocalDist.world, LocalDist.barrier = {}, {}, world, threading.Barrier(world)
        res, errs = {}, []

        def run(rank):
            try:
                s = slab.SlabSolver(LocalDist(rank), rank, world, W, H, halo, lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, 0, first_row=r0))
                r0, r1 = s.local_frame_rows()
                s.set_frames(A[r0:r1], B[r0:r1])
                s.solve(lam, budget, eps=eps)
                a = (s.iterations_done, s.eps_measured) + s.owned_flow()
                s.solve(lam, 20, eps=eps, use_previous=True)
                res[rank] = (s.lo, s.hi, a, (s.iterations_done,) + s.owned_flow())
                s.close()
            except Exception as e:  # surfaced in the main thread
                errs.append((rank, repr(e)))
                LocalDist.barrier.abort()

        ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in ths]
        [t.join(300) for t in ths]
        assert not errs, errs
        for rank in range(world):
            lo, hi, (k, measured, u, v), (kw, uw, vw) = res[rank]
            assert k == i1["iterations_done"] and kw == i2["iterations_done"], (rank, k, i1["iterations_done"], kw, i2["iterations_done"])
            assert measured == (k < budget), (rank, k, measured)
            assert same(u, f1[0][lo:hi]) and same(v, f1[1][lo:hi]) and same(uw, f2[0][lo:hi]) and same(vw, f2[1][lo:hi]), rank
    assert i1["iterations_done"] < 300
