"""CPU tests of the oracle itself.

The reference has no tests or golden vectors for this path, so the oracle is pinned by analytic
known-answer tests, by an independent NumPy restatement and by committed fixtures
(tests/gen_golden.py).  The real cv210.dll cannot be run here; what pins the oracle to it are the reference's own output
pictures (tests/test_reference_pictures.py).
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import hs_numpy
from opticalflowhs_amd import synth

ITER, EPS = 1, 2


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2 ** 31) - a, a)
    b = np.where(b < 0, np.int64(-2 ** 31) - b, b)
    return np.abs(a - b)


def test_golden_files_intact():
    sums = open(os.path.join(GOLDEN, "SHA256SUMS")).read().split("\n")
    n = 0
    for line in sums:
        if not line.strip():
            continue
        digest, name = line.split()
        with open(os.path.join(GOLDEN, name), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == digest, name
        n += 1
    assert n >= 6


def test_k1_identical_frames(oracle):
    A, _ = synth.random_pair(33, 21, seed=3)
    u, v, n, e = oracle.calc_optical_flow_hs(A, A, 0.1, 100, 1e-6, ITER | EPS, return_info=True)
    assert n == 1 and e == 0.0
    assert not u.any() and not v.any()


def test_k2_ramp_closed_form(oracle):
    # A(x) = g*x, B(x) = A(x-1): Ix = g, Iy = 0, It = -g in the interior; k sweeps from zero give
    # u_k = 1 - (1 - g^2 a)^k, v_k = 0 with a = 1/(1/lambda + g^2), away from the borders.
    g, W, H, lam = 2, 64, 40, 0.05
    x = np.arange(W)
    A = np.tile((g * x + 20).astype(np.uint8), (H, 1))
    B = np.tile((g * (x - 1) + 20).astype(np.uint8), (H, 1))
    Ix, Iy, It = oracle.derivatives(A, B)
    assert np.all(Ix[:, 1:-1] == g) and np.all(Iy == 0) and np.all(It == -g)
    a = 1.0 / (1.0 / lam + g * g)
    for k in (1, 3, 10):
        u, v = oracle.calc_optical_flow_hs(A, B, lam, k, term_type=ITER)
        want = 1.0 - (1.0 - g * g * a) ** k
        core = u[k + 1:-(k + 1), k + 1:-(k + 1)]
        assert core.size > 0
        np.testing.assert_allclose(core, want, rtol=2e-6)
        assert not v.any()


def test_k3_symmetries(oracle):
    A, B = synth.smooth_random_pair(40, 28, seed=5, shift=(1, -1))
    u, v = oracle.calc_optical_flow_hs(A, B, 0.5, 30, term_type=ITER)
    # transpose: HS(A^T, B^T) = (v^T, u^T)
    ut, vt = oracle.calc_optical_flow_hs(np.ascontiguousarray(A.T), np.ascontiguousarray(B.T), 0.5, 30, term_type=ITER)
    np.testing.assert_allclose(ut, v.T, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(vt, u.T, atol=2e-6, rtol=1e-5)
    # mirror in x: u -> -u mirrored, v mirrored
    um, vm = oracle.calc_optical_flow_hs(np.ascontiguousarray(A[:, ::-1]), np.ascontiguousarray(B[:, ::-1]), 0.5, 30, term_type=ITER)
    np.testing.assert_allclose(um, -u[:, ::-1], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(vm, v[:, ::-1], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("shape", [(1, 9), (9, 1), (2, 2), (3, 3), (1, 1), (5, 4)])
def test_k4_tiny_images_two_restatements(oracle, shape):
    H, W = shape
    A, B = synth.random_pair(W, H, seed=H * 17 + W)
    for lam in (0.01, 1.0):
        for it in (1, 2, 7):
            u, v = oracle.calc_optical_flow_hs(A, B, lam, it, term_type=ITER)
            u2, v2 = hs_numpy.calc_optical_flow_hs(A, B, lam, it, term_type=ITER)
            um, vm = oracle.calc_optical_flow_hs(A, B, lam, it, term_type=ITER, threads=2)
            assert np.array_equal(u, u2) and np.array_equal(v, v2)
            assert np.array_equal(u, um) and np.array_equal(v, vm)


def test_k5_golden_random(oracle):
    d = np.load(os.path.join(GOLDEN, "k5_random.npz"))
    for (W, H) in ((37, 29), (64, 48)):
        A, B = d["A_%dx%d" % (W, H)], d["B_%dx%d" % (W, H)]
        A2, B2 = synth.random_pair(W, H, seed=W * 1000 + H)
        assert np.array_equal(A, A2) and np.array_equal(B, B2)
        for lam in (0.01, 0.1, 1.0, 10.0):
            for it in (1, 2, 10, 100):
                u, v = oracle.calc_optical_flow_hs(A, B, lam, it, term_type=ITER)
                assert np.array_equal(u, d["u_%dx%d_l%g_i%d" % (W, H, lam, it)])
                assert np.array_equal(v, d["v_%dx%d_l%g_i%d" % (W, H, lam, it)])


def test_golden_numpy_restatement_agrees():
    d = np.load(os.path.join(GOLDEN, "k5_random.npz"))
    A, B = d["A_37x29"], d["B_37x29"]
    for lam in (0.01, 10.0):
        u, v = hs_numpy.calc_optical_flow_hs(A, B, lam, 10, term_type=ITER)
        assert np.array_equal(u, d["u_37x29_l%g_i10" % lam]) and np.array_equal(v, d["v_37x29_l%g_i10" % lam])


def test_eps_termination_golden(oracle):
    d = np.load(os.path.join(GOLDEN, "eps_48x40_l0.002_e1e-3.npz"))
    u, v, n, e = oracle.calc_optical_flow_hs(d["A"], d["B"], 0.002, 500, 1e-3, ITER | EPS, return_info=True)
    assert n == int(d["iters"]) and np.float32(e) == d["eps"]
    assert np.array_equal(u, d["u"]) and np.array_equal(v, d["v"])
    # EPS alone (no ITER) stops at the same sweep; ITER alone runs the whole budget
    u2, v2, n2, _ = oracle.calc_optical_flow_hs(d["A"], d["B"], 0.002, 0, 1e-3, EPS, return_info=True)
    assert n2 == n and np.array_equal(u2, u)


def test_use_previous_continues(oracle):
    A, B = synth.random_pair(31, 23, seed=11)
    u10, v10 = oracle.calc_optical_flow_hs(A, B, 0.3, 10, term_type=ITER)
    u4, v4 = oracle.calc_optical_flow_hs(A, B, 0.3, 4, term_type=ITER)
    u6, v6 = oracle.calc_optical_flow_hs(A, B, 0.3, 6, term_type=ITER, use_previous=True, velx=u4, vely=v4)
    assert np.array_equal(u6, u10) and np.array_equal(v6, v10)


def test_synthetic_translation_recovered(oracle):
    d = np.load(os.path.join(GOLDEN, "synth_256x128_s1_l1_i100.npz"))
    A, B = synth.translating_pair(256, 128, seed=1)
    assert np.array_equal(A, d["A"]) and np.array_equal(B, d["B"])
    u, v = oracle.calc_optical_flow_hs(A, B, 1.0, 100, term_type=ITER, threads=0)
    assert np.array_equal(u, d["u"]) and np.array_equal(v, d["v"])
    # direction and rough magnitude of the known translation (0.75, -0.5)
    assert abs(u[20:-20, 20:-20].mean() - 0.75) < 0.1 and abs(v[20:-20, 20:-20].mean() + 0.5) < 0.1


def test_bunny_golden(oracle):
    from PIL import Image  # PGM decoding only
    fr = [oracle.box_blur3(np.asarray(Image.open(os.path.join(GOLDEN, "bunny_%d_gray.pgm" % i)))) for i in (1, 2)]
    assert fr[0].shape == (240, 424)
    d = np.load(os.path.join(GOLDEN, "bunny_flow_l1_i50.npz"))
    u, v, n, _ = oracle.calc_optical_flow_hs(fr[0], fr[1], 1.0, 50, 1e-6, ITER | EPS, threads=0, return_info=True)
    assert n == int(d["iters"]) == 50
    assert np.array_equal(u, d["u"]) and np.array_equal(v, d["v"])


def test_argument_errors(oracle):
    A, B = synth.random_pair(8, 8, seed=1)
    with pytest.raises(ValueError):
        oracle.calc_optical_flow_hs(A, B[:4], 1.0, 3)
    with pytest.raises(ValueError):
        oracle.calc_optical_flow_hs(A.astype(np.float32), B, 1.0, 3)
    with pytest.raises(ValueError):
        oracle.calc_optical_flow_hs(A, B, 1.0, 0, term_type=ITER)


def test_preproc_oracle(oracle):
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    g = oracle.bgr2gray(bgr)
    c = bgr.astype(np.int64)
    want = (1868 * c[..., 0] + 9617 * c[..., 1] + 4899 * c[..., 2] + 8192) >> 14
    assert np.array_equal(g, want.astype(np.uint8))
    assert oracle.bgr2gray(np.full((2, 2, 3), 255, np.uint8)).max() == 255
    b = oracle.box_blur3(g)
    p = np.pad(g.astype(np.int64), 1, mode="edge")
    s = sum(p[dy:dy + 9, dx:dx + 13] for dy in range(3) for dx in range(3))
    assert np.array_equal(b, np.rint(s / 9.0).astype(np.uint8))


def test_classic_oracle_basics(oracle):
    A, B = synth.smooth_random_pair(24, 18, seed=9, shift=(1, 0))
    Ex, Ey, Et = oracle.classic_derivatives(A, B)
    a = np.pad(A.astype(np.float64), ((0, 1), (0, 1)), mode="edge")
    b = np.pad(B.astype(np.float64), ((0, 1), (0, 1)), mode="edge")
    ex = 0.25 * (a[:-1, 1:] - a[:-1, :-1] + a[1:, 1:] - a[1:, :-1] + b[:-1, 1:] - b[:-1, :-1] + b[1:, 1:] - b[1:, :-1])
    et = 0.25 * (b[:-1, :-1] - a[:-1, :-1] + b[:-1, 1:] - a[:-1, 1:] + b[1:, :-1] - a[1:, :-1] + b[1:, 1:] - a[1:, 1:])
    assert np.array_equal(Ex, ex.astype(np.float32)) and np.array_equal(Et, et.astype(np.float32))
    u, v = oracle.classic_flow(A, B, 15.0, 20)
    assert np.isfinite(u).all() and np.abs(v).max() > 0  # the v update is restored
    u0, v0 = oracle.classic_flow(A, A, 15.0, 5)
    assert not u0.any() and not v0.any()


def test_oracle_under_sanitizers(tmp_path):
    """Every entry point of the C oracle on small and degenerate sizes with exact-size heap buffers,
    built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_san")
    srcs = [os.path.join(root, "tests", "oracle_sanitize_main.c")] + \
           [os.path.join(root, "oracle", f) for f in ("hs_cv_oracle.c", "hs_classic_oracle.c", "hs_preproc_oracle.c")]
    r = subprocess.run([gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-fopenmp",
                        "-o", exe] + srcs + ["-lm"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="3"))
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.returncode, r.stdout, r.stderr[-3000:])
