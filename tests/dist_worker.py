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

    def set_frames(self, prev, curr):
        self.prev, self.curr = np.ascontiguousarray(prev), np.ascontiguousarray(curr)

    def sweep(self, n, lam, first):
        self.u, self.v = hs_oracle.calc_optical_flow_hs(self.prev, self.curr, lam, n, term_type=hs_oracle.TERMCRIT_ITER,
                                                       use_previous=not first, velx=self.u, vely=self.v)

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
    n_ex = s.solve(0.7, iters)
    u, v = s.owned_flow()
    np.savez(os.path.join(out, "rank%d.npz" % rank), u=u, v=v, lo=s.lo, hi=s.hi, n_ex=n_ex,
             pairs=np.array(slab.shard_pairs(11, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
