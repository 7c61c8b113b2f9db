"""Several GPUs from one host process (include/hsflow.h: hsflow_multi_*, hsflow_slab_*).  The GPU box has one card, so
the slabs / workers share it (a device may be listed more than once): the device-to-device halo exchange then is an
ordinary copy instead of a peer copy, everything else is the code a multi-GPU node runs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from opticalflowhs_amd import synth

pytestmark = pytest.mark.gpu

ITER, EPS = 1, 2


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
def test_c_host_slabs_and_pairs(hs, gpu_ok, tmp_path, devices):
    """tests/multi_host.c: a C99 program on the public header, linked against libhsflow.so."""
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libdir = os.path.dirname(hs._lib.LIB_PATH)
    exe = str(tmp_path / "multi_host")
    r = subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "multi_host.c"),
                        "-o", exe, "-L", libdir, "-lhsflow", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the library binds to the HIP runtime PyTorch ships (same soname as /opt/rocm's): one runtime per process
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + tlib + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, devices], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "slab ok" in r.stdout and "multi ok" in r.stdout


@pytest.mark.parametrize("devices,halo,iters,shape", [((0,), 8, 30, (300, 512)), ((0, 0), 8, 30, (300, 512)), ((0, 0, 0), 12, 40, (203, 700)),
                                                      ((0, 0, 0, 0), 16, 100, (517, 1300))])
def test_slab_frame_is_the_whole_frame_solve(hs, oracle, gpu_ok, devices, halo, iters, shape):
    H, W = shape
    A, B = synth.translating_pair(W, H, seed=3)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ctx.solve(lam=0.7, max_iter=iters, term_type=ITER)
        uo, vo = ctx.flow()
    with hs.SlabFrame(W, H, devices=devices, halo=halo) as s:
        rows = s.rows()
        assert rows[0][0] == 0 and rows[-1][1] == H and all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
        s.set_frames(A, B)
        for rep in range(2):   # a second solve on the same object starts from zero again
            n_ex = s.solve(lam=0.7, max_iter=iters)
            assert n_ex == (-(-iters // halo) - 1 if len(devices) > 1 else 0)
            u, v = s.flow()
            assert np.array_equal(u, uo) and np.array_equal(v, vo), rep
    if len(devices) == 3:
        ur, vr = oracle.calc_optical_flow_hs(A, B, 0.7, iters, term_type=ITER)
        assert np.sqrt(np.mean((u.astype(np.float64) - ur) ** 2)) <= 1e-4 and np.sqrt(np.mean((v.astype(np.float64) - vr) ** 2)) <= 1e-4


def test_multi_pairs_round_robin(hs, gpu_ok):
    W, H, it, n = 320, 200, 25, 9
    pairs = [synth.translating_pair(W, H, seed=500 + i, dx=0.3 * i, dy=-0.4) if i % 2 else synth.random_pair(W, H, seed=500 + i) for i in range(n)]
    ref = []
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        for A, B in pairs:
            ctx.set_frames(A, B)
            ctx.solve(lam=0.5, max_iter=it, term_type=ITER)
            ref.append(ctx.flow())
    for devices in ((0,), (0, 0, 0)):
        with hs.MultiPairs(W, H, devices=devices, depth=2) as mp:
            bufs = []
            for A, B in pairs:
                a, b = hs.pinned_empty((H, W), np.uint8), hs.pinned_empty((H, W), np.uint8)
                a[...], b[...] = A, B
                u, v = hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)
                u.fill(np.nan)
                v.fill(np.nan)
                mp.submit(a, b, u, v, lam=0.5, max_iter=it, term_type=ITER | EPS, epsilon=float(np.float32(1e-6)))
                bufs.append((u, v))
            mp.wait(n - 1)
            mp.drain()
            for i, (u, v) in enumerate(bufs):
                assert np.array_equal(u, ref[i][0]) and np.array_equal(v, ref[i][1]), (devices, i)
    with pytest.raises(hs.HsflowError):
        hs.MultiPairs(W, H, devices=(99,))
    with pytest.raises(hs.HsflowError):
        hs.SlabFrame(W, 20, devices=(0, 0, 0), halo=16)


def same_flow(x, y):
    """Bit for bit, except where both values lie below 1e-30: inside a launch the strip kernels carry 4^k u, so flow that would
    be denormal (< 1.2e-38) keeps bits that depend on where the launch boundaries fall (chunks of `halo` sweeps against
    launches of fuse_steps; DESIGN.md 4.1 "Scaled state"), and such an addend can still decide the rounding of sums up to
    about 1e-32.  Only synthetic flat frames get there."""
    return bool(np.all((x == y) | ((np.abs(x) < 1e-30) & (np.abs(y) < 1e-30))))


@pytest.mark.parametrize("devices,overlapped", [((0, 0), False), ((0, 0, 0), False), ((0,), True), ((0, 0), True)])
def test_slab_frame_iter_eps_stops_where_the_whole_frame_solve_stops(hs, gpu_ok, devices, overlapped):
    """ITER|EPS over row slabs (the reference's own criteria, /root/reference OpticalFlowOpenCV.cpp:29): the frame's Eps is
    the maximum over the slabs' owned rows.  Budgets that run out (some slab's witness vouches for every chunk), stops
    inside the first, a middle and the last chunk, a stop exactly at a chunk's end, and warm starts -- flow and stopping
    sweep must be those of the one-context solve, bit for bit."""
    W, H, halo = 600, 260, 12
    flat_a = np.full((H, W), 90, np.uint8)
    flat_b = flat_a.copy()
    flat_a[50:200, 100:480] = 120
    flat_b[50:200, 100:480] = 121
    moving = synth.translating_pair(W, H, seed=21)
    cases = [(moving, 1.0, 70, float(np.float32(1e-6))),     # the budget runs out, witnessed
             ((flat_a, flat_b), 1e-3, 400, 1e-4),            # converges in a middle chunk
             ((flat_a, flat_b), 1e-3, 400, 1e-2),            # ... in the first chunk
             ((flat_a, flat_b), 1e-3, 400, 3e-5),            # ... late
             ((flat_a, flat_a), 1.0, 50, 1e-6)]              # identical frames: Eps = 0 in sweep 1
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx, hs.SlabFrame(W, H, devices=devices, halo=halo, overlapped=overlapped) as s:
        stops = []
        for (A, B), lam, budget, eps in cases:
            ctx.set_frames(A, B)
            s.set_frames(A, B)
            i = ctx.solve(lam=lam, max_iter=budget, term_type=ITER | EPS, epsilon=eps)
            uo, vo = ctx.flow()
            s.solve(lam=lam, max_iter=budget, term_type=ITER | EPS, epsilon=eps)
            u, v = s.flow()
            assert s.iterations_done() == i["iterations_done"], (lam, budget, eps, s.iterations_done(), i["iterations_done"])
            assert same_flow(u, uo) and same_flow(v, vo), (lam, budget, eps)
            stops.append(i["iterations_done"])
            # a warm start from that flow, again ITER|EPS
            i2 = ctx.solve(lam=lam, max_iter=25, term_type=ITER | EPS, epsilon=eps, use_previous=True)
            uo, vo = ctx.flow()
            s.solve(lam=lam, max_iter=25, term_type=ITER | EPS, epsilon=eps, use_previous=True)
            u, v = s.flow()
            assert s.iterations_done() == i2["iterations_done"] and same_flow(u, uo) and same_flow(v, vo), (lam, budget, eps, "warm")
        assert stops[0] == 70 and 1 < stops[1] < 400 and stops[2] <= halo and stops[4] == 1, stops
        # an exact stop at a chunk's end, whatever sweep that is: epsilon just above that sweep's Eps
        (A, B), lam = cases[1][0], 1e-3
        ctx.set_frames(A, B)
        s.set_frames(A, B)
        eps_k = ctx.solve_probe(lam=lam, max_iter=3 * halo, term_type=ITER)
        eps = float(np.nextafter(eps_k[2 * halo - 1], np.float32(np.inf)))
        if np.all(eps_k[:2 * halo - 1] >= eps):   # (Eps falls monotonically on this pair: the stop is sweep 2 * halo)
            i = ctx.solve(lam=lam, max_iter=400, term_type=ITER | EPS, epsilon=eps)
            uo, vo = ctx.flow()
            s.solve(lam=lam, max_iter=400, term_type=ITER | EPS, epsilon=eps)
            u, v = s.flow()
            assert i["iterations_done"] == 2 * halo == s.iterations_done() and same_flow(u, uo) and same_flow(v, vo)
