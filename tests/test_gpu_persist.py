"""The persistent launch (HSFLOW_KERNEL_PERSIST: one launch per solve, workgroups keep their tile in registers across
phases and swap halos through HBM behind per-tile phase counters; replaces the host loop of
/root/reference's HSOpticalFlowOpenCL.cpp:748-752 inside one kernel).  Bar: bit-identical to the launch-per-fuse_steps
strip kernel (itself bit-identical to the one-sweep kernel and within 1e-4 RMS of the oracle, test_gpu_parity.py),
every refusal explicit, a timed-out wait repeated launch by launch."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from opticalflowhs_amd import synth

pytestmark = pytest.mark.gpu
ITER, EPS = 1, 2
EPS6 = float(np.float32(1e-6))


def solve(hs, ctx, sync, **kw):
    if sync:
        info = ctx.solve(**kw)
    else:
        ctx.solve_async(**kw)
        ctx.synchronize()
        info = ctx.info()
    u, v = ctx.flow()
    return u, v, info


@pytest.mark.parametrize("shape,iters,T", [((1920, 1080), 100, 0), ((1920, 1080), 93, 20), ((1920, 1080), 41, 16), ((1280, 720), 60, 12),
                                          ((1024, 200), 35, 10), ((256, 80), 50, 8)])
def test_persistent_launch_is_bit_identical_to_the_strip_kernel(hs, gpu_ok, shape, iters, T):
    W, H = shape
    A, B = synth.translating_pair(W, H, seed=7)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        u0, v0, i0 = solve(hs, ctx, True, lam=1.0, max_iter=iters, term_type=ITER, kernel=hs.KERNEL_STRIP, fuse_steps=T, strip_rows=5)
        Tp = i0["fuse_steps"]
        for sync, graph, tt in ((True, False, ITER), (False, True, ITER), (False, False, ITER | EPS), (False, True, ITER | EPS)):
            try:
                u, v, i = solve(hs, ctx, sync, lam=1.0, max_iter=iters, term_type=tt, epsilon=EPS6, kernel=hs.KERNEL_PERSIST,
                                fuse_steps=Tp, strip_rows=5, use_graph=graph)
            except hs.HsflowError as e:
                # shapes whose plan has more tiles than CUs, a halo beyond the neighbouring tile, or a single phase
                assert e.status == hs._lib.E_SIZE and "HSFLOW_KERNEL_PERSIST" in str(e), (shape, iters, T, e)
                assert shape != (1920, 1080) or T == 16, e
                continue
            assert i["persistent"] == -(-iters // Tp) >= 2 and i["jacobi_launches"] == 1 and i["kernel"] == hs.KERNEL_STRIP, i
            assert i["iterations_done"] == iters and i["eps_rerun"] == 0, i
            assert np.array_equal(u, u0) and np.array_equal(v, v0), (shape, iters, Tp, sync, graph, tt)
            if tt & EPS:  # last_eps of an asynchronous solve is measured on demand: the last phase again, from the third buffer
                s = ctx.solve(lam=1.0, max_iter=iters, term_type=tt, epsilon=EPS6, kernel=hs.KERNEL_STRIP, fuse_steps=Tp, strip_rows=5)
                assert i["last_eps"] == s["last_eps"] > 0, (i["last_eps"], s["last_eps"])


def test_persistent_launch_warm_start_and_repeats(hs, gpu_ok):
    """use_previous (the starting flow stays intact: the phases write the other two buffers), solves back to back on
    the same counters, and a change of the grid (another fuse_steps) in between."""
    W, H = 1920, 1080
    A, B = synth.translating_pair(W, H, seed=3)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ref = []
        for k, (it, prev, T) in enumerate(((40, False, 20), (30, True, 20), (45, True, 20), (60, False, 14), (25, True, 20))):
            ctx.solve(lam=2.0, max_iter=it, term_type=ITER, kernel=hs.KERNEL_STRIP, fuse_steps=T, use_previous=prev)
            ref.append(ctx.flow())
        for k, (it, prev, T) in enumerate(((40, False, 20), (30, True, 20), (45, True, 20), (60, False, 14), (25, True, 20))):
            ctx.solve_async(lam=2.0, max_iter=it, term_type=ITER, kernel=hs.KERNEL_PERSIST, fuse_steps=T, use_previous=prev, use_graph=bool(k & 1))
            u, v = ctx.flow()
            assert ctx.info()["persistent"] >= 2
            assert np.array_equal(u, ref[k][0]) and np.array_equal(v, ref[k][1]), k


def test_persistent_launch_early_stop_is_found(hs, oracle, gpu_ok):
    """ITER|EPS on a pair whose iteration converges inside the budget: the witness phases cannot prove "no early stop",
    the exact pass runs and the stopping sweep is the oracle's."""
    W, H = 512, 160
    A = np.full((H, W), 90, np.uint8)
    B = A.copy()
    A[40:120, 100:400] = 120   # one flat patch, identical in both frames but for a one-level step
    B[40:120, 100:400] = 121
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        s = ctx.solve(lam=1e-3, max_iter=400, term_type=ITER | EPS, epsilon=1e-4, kernel=hs.KERNEL_STRIP, fuse_steps=16, strip_rows=5)
        us, vs = ctx.flow()
        assert 1 < s["iterations_done"] < 400, s
        ctx.solve_async(lam=1e-3, max_iter=400, term_type=ITER | EPS, epsilon=1e-4, kernel=hs.KERNEL_PERSIST, fuse_steps=16, strip_rows=5)
        ctx.synchronize()
        i = ctx.info()
        u, v = ctx.flow()
    assert i["iterations_done"] == s["iterations_done"] and i["eps_rerun"] == 1, (i, s)
    assert np.array_equal(u, us) and np.array_equal(v, vs)
    uo, vo, k, _ = oracle.calc_optical_flow_hs(A, B, 1e-3, 400, term_type=ITER | EPS, epsilon=1e-4, return_info=True)
    assert abs(k - i["iterations_done"]) <= 1, (k, i)


def test_persistent_launch_refusals(hs, gpu_ok):
    A, B = synth.translating_pair(1920, 1080, seed=1)
    with hs.HSFlow(1920, 1080, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        kw = dict(lam=1.0, max_iter=100, term_type=ITER, kernel=hs.KERNEL_PERSIST)
        assert ctx.solve(**kw)["persistent"] == 5
        # synchronous ITER|EPS measures its last sweep; the folded kernel, other row counts and a single phase have no persistent form
        for bad in (dict(term_type=ITER | EPS), dict(strip_rows=4), dict(max_iter=20, fuse_steps=20), dict(term_type=EPS)):
            with pytest.raises(hs.HsflowError) as e:
                ctx.solve(**dict(kw, **bad))
            assert e.value.status == hs._lib.E_SIZE and "HSFLOW_KERNEL_PERSIST" in str(e.value), (bad, e.value)
        # a second context alive on the device: two persistent grids could hold part of the CUs each and starve
        with hs.HSFlow(64, 64, 1, own_stream=True):
            with pytest.raises(hs.HsflowError) as e:
                ctx.solve(**kw)
            assert "another context" in str(e.value)
        assert ctx.solve(**kw)["persistent"] == 5
        # AUTO does not take it (a phase boundary costs what a kernel boundary costs: DESIGN.md 4.4)
        assert ctx.solve(lam=1.0, max_iter=100, term_type=ITER)["persistent"] == 0
    for (W, H) in ((3840, 2160), (1918, 1080), (200, 1080)):   # more tiles than CUs; width not a multiple of 4; narrower than a region
        with hs.HSFlow(W, H, 1, own_stream=True) as c2:
            c2.set_frames(*synth.translating_pair(W, H, seed=2))
            with pytest.raises(hs.HsflowError) as e:
                c2.solve(lam=1.0, max_iter=100, term_type=ITER, kernel=hs.KERNEL_PERSIST)
            assert e.value.status == hs._lib.E_SIZE


def test_persistent_launch_timeout_falls_back(gpu_ok):
    """HSFLOW_PERSIST_WAIT_TICKS=0 makes every wait that is not already satisfied give up: the workgroups write the error
    word and leave, the host sees it when the stream has drained, repeats the solve launch by launch and does not use
    the persistent launch on that context again.  (Fresh process: the knob is read once.)"""
    code = r'''
import numpy as np, opticalflowhs_amd as hs
from opticalflowhs_amd import synth
A, B = synth.translating_pair(1920, 1080, seed=5)
with hs.HSFlow(1920, 1080, 1, own_stream=True) as ctx:
    ctx.set_frames(A, B)
    ctx.solve(lam=1.0, max_iter=100, term_type=1, kernel=hs.KERNEL_STRIP)
    u0, v0 = ctx.flow()
    i = ctx.solve(lam=1.0, max_iter=100, term_type=1, kernel=hs.KERNEL_PERSIST)          # synchronous: repeated at once
    u, v = ctx.flow()
    assert i["persistent"] == 0 and i["iterations_done"] == 100 and i["jacobi_launches"] == 5, i
    assert np.array_equal(u, u0) and np.array_equal(v, v0)
    try:
        ctx.solve(lam=1.0, max_iter=100, term_type=1, kernel=hs.KERNEL_PERSIST)
        raise SystemExit("a context whose persistent launch timed out must refuse the next one")
    except hs.HsflowError as e:
        assert "timed out" in str(e), e
with hs.HSFlow(1920, 1080, 1, own_stream=True) as ctx:                                     # asynchronous ITER|EPS: settled by the exact pass
    ctx.set_frames(A, B)
    ctx.solve_async(lam=1.0, max_iter=100, term_type=3, epsilon=1e-6, kernel=hs.KERNEL_PERSIST)
    ctx.synchronize()
    i = ctx.info()
    u, v = ctx.flow()
    assert i["iterations_done"] == 100 and i["eps_rerun"] == 1, i
    assert np.array_equal(u, u0) and np.array_equal(v, v0)
with hs.HSFlow(1920, 1080, 1, own_stream=True) as ctx:                                     # asynchronous ITER: reported when the stream is drained
    ctx.set_frames(A, B)
    ctx.solve_async(lam=1.0, max_iter=100, term_type=1, kernel=hs.KERNEL_PERSIST)
    try:
        ctx.synchronize()
        raise SystemExit("the timed-out launch went unnoticed")
    except hs.HsflowError as e:
        assert e.status == hs._lib.E_DEVICE and "timed out" in str(e), e
    ctx.solve(lam=1.0, max_iter=100, term_type=1)
    u, v = ctx.flow()
    assert np.array_equal(u, u0) and np.array_equal(v, v0)
print("FALLBACK-OK")
'''
    env = dict(os.environ, HSFLOW_PERSIST_WAIT_TICKS="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "FALLBACK-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
