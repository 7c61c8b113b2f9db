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
