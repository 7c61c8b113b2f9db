"""Worker for tests/test_dist.py: one process per rank, gloo backend, CPU tensors.

The slab driver (opticalflowhs_amd/slab.py) is exercised with a backend that wraps the CPU oracle
(test infrastructure): what is under test here is the partitioning and the halo exchange, i.e.
that chunked sweeps + row swaps reproduce the single-domain solve bit for bit.
usage: python -m torch.distributed.run --nproc-per-node N tests/dist_worker.py W H HALO ITERS OUT [overlap]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticalflowhs_amd import slab, synth  # noqa: E402
from oracle import hs_oracle  # noqa: E402


class OracleSlabBackend(object):
    def __init__(self, width, height):
        self.width, self.height = width, height
        self.u = np.zeros((height, width), np.float32)
        self.v = np.zeros((height, width), np.float32)
        self.win = (0, height)
        self._proven = False

    def set_frames(self, prev, curr):
        self.prev, self.curr = np.ascontiguousarray(prev), np.ascontiguousarray(curr)

    def _one(self, lam, from_zero):
        """One sweep; returns its Eps over the rows of set_eps_rows (fp32 differences, like the original's)."""
        u, v = hs_oracle.calc_optical_flow_hs(self.prev, self.curr, lam, 1, term_type=hs_oracle.TERMCRIT_ITER,
                                              use_previous=not from_zero, velx=self.u, vely=self.v)
        u0 = np.zeros_like(u) if from_zero else self.u
        v0 = np.zeros_like(v) if from_zero else self.v
        a, b = self.win
        e = max(float(np.max(np.abs(u0[a:b] - u[a:b]))), float(np.max(np.abs(v0[a:b] - v[a:b]))))
        self.u, self.v = u, v
        return np.float32(e)

    def sweep(self, n, lam, first, from_zero=None, eps=None):
        from_zero = first if from_zero is None else from_zero
        if eps is None:
            self.u, self.v = hs_oracle.calc_optical_flow_hs(self.prev, self.curr, lam, n, term_type=hs_oracle.TERMCRIT_ITER,
                                                           use_previous=not from_zero, velx=self.u, vely=self.v)
            return
        # "witness": here simply the truth -- did Eps stay >= eps in every sweep of the chunk, over my rows?
        e = [self._one(lam, from_zero and k == 0) for k in range(n)]
        self._proven = all(float(x) >= eps for x in e)

    def set_eps_rows(self, first_row, rows):
        self.win = (first_row, first_row + rows)

    def verdict(self):
        return self._proven

    def probe(self, n, lam, first, from_zero=None):
        from_zero = first if from_zero is None else from_zero
        return np.array([self._one(lam, from_zero and k == 0) for k in range(n)], np.float32)

    def solve_whole(self, lam, iters, eps, use_previous):
        tt = hs_oracle.TERMCRIT_ITER if eps is None else hs_oracle.TERMCRIT_ITER | hs_oracle.TERMCRIT_EPS
        self.u, self.v, it, _ = hs_oracle.calc_optical_flow_hs(self.prev, self.curr, lam, iters, epsilon=eps if eps is not None else 1e-6, term_type=tt,
                                                              use_previous=use_previous, velx=self.u, vely=self.v, return_info=True)
        return it, False

    def save(self):
        return self.u.copy(), self.v.copy()

    def restore(self, saved):
        self.u, self.v = saved[0].copy(), saved[1].copy()

    def new_rows(self, nrows):
        return torch.empty((nrows, self.width), dtype=torch.float32), torch.empty((nrows, self.width), dtype=torch.float32)

    def get_rows(self, row0, u, v):
        u.copy_(torch.from_numpy(self.u[row0:row0 + u.shape[0]]))
        v.copy_(torch.from_numpy(self.v[row0:row0 + v.shape[0]]))

    def put_rows(self, row0, u, v):
        self.u[row0:row0 + u.shape[0]] = u.numpy()
        self.v[row0:row0 + v.shape[0]] = v.numpy()

    def flow(self):
        return self.u, self.v

    def close(self):
        pass

    # everything above is synchronous: the ordering hooks of the overlapped driver are no-ops
    def activate(self):
        import contextlib
        return contextlib.nullcontext()

    def record(self):
        return None

    def wait_event(self, ev):
        pass


def main():
    W, H, halo, iters = (int(x) for x in sys.argv[1:5])
    out = sys.argv[5]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    overlapped = len(sys.argv) > 6 and sys.argv[6] == "overlap"
    cls = slab.OverlappedSlabSolver if overlapped else slab.SlabSolver
    s = cls(dist, rank, world, W, H, halo, OracleSlabBackend)
    r0, r1 = s.local_frame_rows()
    A, B = synth.translating_pair(W, H, seed=3, row0=r0, rows=r1 - r0)  # each rank makes only its rows
    s.set_frames(A, B)
    extra = {}
    if len(sys.argv) > 6 and sys.argv[6].startswith("eps="):
        # ITER|EPS over the slabs, then a warm start: "eps=<epsilon>,<lambda>"
        eps, lam = (float(x) for x in sys.argv[6][4:].split(","))
        n_ex = s.solve(lam, iters, eps=eps)
        extra = dict(done=s.iterations_done, measured=s.eps_measured)
        u, v = s.owned_flow()
        s.solve(lam, 7, eps=eps, use_previous=True)
        uw, vw = s.owned_flow()
        extra.update(uw=uw, vw=vw, done_warm=s.iterations_done)
    else:
        n_ex = s.solve(0.7, iters)
        u, v = s.owned_flow()
    np.savez(os.path.join(out, "rank%d.npz" % rank), u=u, v=v, lo=s.lo, hi=s.hi, n_ex=n_ex,
             pairs=np.array(slab.shard_pairs(11, world, rank)), **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
