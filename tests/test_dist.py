"""Multi-process tests of the N > 1 paths on CPU (gloo, world_size 2 and 3).

Independent pairs shard with no collective (only the index arithmetic is testable without a GPU);
the row-slab mode has a real exchange step, checked here end to end against the single-domain
oracle result, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from opticalflowhs_amd import slab, synth


def test_slab_partition_arithmetic():
    for H in (7, 64, 1080, 16384):
        for world in (1, 2, 3, 8):
            rows = [slab.slab_rows(H, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in rows) - min(b - a for a, b in rows) <= 1
    lo, hi, top, bot = slab.slab_extent(100, 4, 0, 8)
    assert (lo, hi, top, bot) == (0, 25, 0, 8)
    lo, hi, top, bot = slab.slab_extent(100, 4, 3, 8)
    assert (lo, hi, top, bot) == (75, 100, 8, 0)
    assert slab.chunks(100, 16) == [16] * 6 + [4] and slab.chunks(8, 8) == [8] and slab.chunks(5, 8) == [5]
    # independent pairs: every pair exactly once, round robin
    got = sorted(i for r in range(8) for i in slab.shard_pairs(512, 8, r))
    assert got == list(range(512)) and len(slab.shard_pairs(512, 8, 3)) == 64
    with pytest.raises(ValueError):
        slab.SlabSolver(None, 0, 4, 32, 20, 8, lambda w, h: None)  # 5-row slabs, halo 8


@pytest.mark.parametrize("world,halo,iters,mode", [(2, 4, 10, "plain"), (2, 8, 8, "plain"), (3, 5, 23, "plain"),
                                                   (2, 4, 13, "overlap"), (3, 5, 23, "overlap"), (1, 6, 20, "overlap")])
def test_slab_exchange_matches_single_domain(tmp_path, oracle, world, halo, iters, mode):
    """mode "overlap": two sub-slabs per rank worked alternately (OverlappedSlabSolver)."""
    W, H = 96, 61
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world * 10 + halo + (100 if mode == "overlap" else 0)),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(W), str(H), str(halo), str(iters), str(tmp_path)] + \
          (["overlap"] if mode == "overlap" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    A, B = synth.translating_pair(W, H, seed=3)
    uo, vo = oracle.calc_optical_flow_hs(A, B, 0.7, iters, term_type=1)
    u = np.zeros_like(uo)
    v = np.zeros_like(vo)
    pairs = []
    for rank in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        u[int(d["lo"]):int(d["hi"])] = d["u"]
        v[int(d["lo"]):int(d["hi"])] = d["v"]
        assert int(d["n_ex"]) == len(slab.chunks(iters, halo)) - 1
        pairs += list(d["pairs"])
    assert np.array_equal(u, uo) and np.array_equal(v, vo)
    assert sorted(pairs) == list(range(11))


@pytest.mark.parametrize("world,halo,iters,eps,lam", [(2, 4, 30, 1e-6, 0.7), (2, 5, 60, 2e-2, 0.7), (3, 4, 60, 5e-3, 0.7), (2, 6, 40, 0.3, 0.7), (1, 4, 60, 2e-2, 0.7)])
def test_slab_iter_eps_matches_single_domain(tmp_path, oracle, world, halo, iters, eps, lam):
    """ITER|EPS over the ranks (gloo, all_reduce MAX of the witness verdicts and of the per-sweep Eps): the stopping sweep
    and the flow of the single-domain oracle solve, and of a warm start after it, bit for bit."""
    W, H = 96, 61
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world * 10 + halo),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(W), str(H), str(halo), str(iters), str(tmp_path), "eps=%r,%r" % (eps, lam)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    A, B = synth.translating_pair(W, H, seed=3)
    uo, vo, k, _ = oracle.calc_optical_flow_hs(A, B, lam, iters, epsilon=eps, term_type=3, return_info=True)
    uw, vw, kw, _ = oracle.calc_optical_flow_hs(A, B, lam, 7, epsilon=eps, term_type=3, use_previous=True, velx=uo, vely=vo, return_info=True)
    u, v, u2, v2 = (np.zeros_like(uo) for _ in range(4))
    for rank in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        lo, hi = int(d["lo"]), int(d["hi"])
        u[lo:hi], v[lo:hi], u2[lo:hi], v2[lo:hi] = d["u"], d["v"], d["uw"], d["vw"]
        assert int(d["done"]) == k and int(d["done_warm"]) == kw, (rank, int(d["done"]), k, int(d["done_warm"]), kw)
        assert bool(d["measured"]) == (k < iters and world > 1) or world == 1 or k == iters
    assert np.array_equal(u, uo) and np.array_equal(v, vo)
    assert np.array_equal(u2, uw) and np.array_equal(v2, vw)
