"""GPU tests of the rows either side of the hot path: the pre-processing kernels
(BGR->gray, 3x3 box blur), streaming frames, and the C++ drop-in (class + CLI) built by `make host`."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from opticalflowhs_amd import synth

pytestmark = pytest.mark.gpu
ITER, EPS = 1, 2


def rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def write_pnm(path, img):
    with open(path, "wb") as f:
        if img.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        else:
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())


def read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(x) for x in f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("shape", [(9, 13), (48, 64), (61, 203), (1, 5), (5, 1), (240, 424)])
def test_preprocessing_bit_exact(hs, oracle, gpu_ok, shape):
    H, W = shape
    rng = np.random.default_rng(H * 7 + W)
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames_bgr(a, b, blur=False)
        ga, gb = ctx.frames()
        assert np.array_equal(ga, oracle.bgr2gray(a)) and np.array_equal(gb, oracle.bgr2gray(b))
        ctx.set_frames_bgr(a, b, blur=True)
        fa, fb = ctx.frames()
        assert np.array_equal(fa, oracle.box_blur3(oracle.bgr2gray(a)))
        assert np.array_equal(fb, oracle.box_blur3(oracle.bgr2gray(b)))
        ctx.set_frames_gray_blur(ga, gb)
        fa2, fb2 = ctx.frames()
        assert np.array_equal(fa2, fa) and np.array_equal(fb2, fb)


def test_bunny_end_to_end_from_gray(hs, gpu_ok):
    """BASELINE config C1 as the reference's CPU route runs it: blur inside, lambda 1, 50 iterations,
    ITER|EPS with eps 1e-6 (OpticalFlowOpenCV.cpp:26-30), here entirely on the GPU."""
    from PIL import Image  # PGM decoding only
    g = [np.asarray(Image.open(os.path.join(GOLDEN, "bunny_%d_gray.pgm" % i))) for i in (1, 2)]
    d = np.load(os.path.join(GOLDEN, "bunny_flow_l1_i50.npz"))
    with hs.HSFlow(424, 240, 1, own_stream=True) as ctx:
        ctx.set_frames_gray_blur(g[0], g[1])
        info = ctx.solve(lam=1.0, max_iter=50, epsilon=float(np.float32(1e-6)), term_type=ITER | EPS)
        u, v = ctx.flow()
    assert info["iterations_done"] == 50
    assert rms(u, d["u"]) <= 1e-4 and rms(v, d["v"]) <= 1e-4


def test_streaming_push_frame(hs, oracle, gpu_ok):
    W, H = 160, 96
    frames = [synth.translating_pair(W, H, seed=5, dx=0.5 * k, dy=0.25 * k)[1] for k in range(4)]
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(frames[0], frames[1])
        for k in range(1, 4):
            if k > 1:
                ctx.push_frame(frames[k])          # frame k-1 stays on the device as "previous"
            ctx.solve(lam=1.0, max_iter=25, term_type=ITER)
            u, v = ctx.flow()
            uo, vo = oracle.calc_optical_flow_hs(frames[k - 1], frames[k], 1.0, 25, term_type=ITER)
            assert rms(u, uo) <= 1e-4 and rms(v, vo) <= 1e-4
            a, b = ctx.frames()
            assert np.array_equal(a, frames[k - 1]) and np.array_equal(b, frames[k])


def test_reuse_derivatives_flag(hs, gpu_ok):
    A, B = synth.random_pair(90, 50, seed=3)
    with hs.HSFlow(90, 50, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ctx.solve(lam=0.4, max_iter=12, term_type=ITER)
        u12, v12 = ctx.flow()
        ctx.solve(lam=0.4, max_iter=5, term_type=ITER)
        ctx.solve(lam=0.4, max_iter=7, term_type=ITER, use_previous=True, reuse_derivatives=True)
        u, v = ctx.flow()
        assert np.array_equal(u, u12) and np.array_equal(v, v12)


@pytest.mark.parametrize("shape", [(18, 24), (61, 203), (1, 7), (7, 1), (2, 2), (240, 424)])
def test_classic_mode_bit_exact(hs, oracle, gpu_ok, shape):
    """Kernels.cl semantics with the v update restored, against its oracle."""
    H, W = shape
    A, B = synth.smooth_random_pair(W, H, seed=H + W, shift=(1, 0)) if min(H, W) > 2 else synth.random_pair(W, H, seed=3)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for alpha, it in ((15.0, 1), (15.0, 20), (1.0, 7)):
            info = ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=it, term_type=ITER)
            assert info["iterations_done"] == it
            u, v = ctx.flow()
            uo, vo = oracle.classic_flow(A, B, alpha, it)
            assert np.array_equal(u, uo) and np.array_equal(v, vo), (alpha, it)
        ex, ey, et = ctx.derivatives()
        Ex, Ey, Et = oracle.classic_derivatives(A, B)
        assert np.array_equal(ex, Ex) and np.array_equal(ey, Ey) and np.array_equal(et, Et)
        # continuing from the current flow, and switching back to the CV discretisation
        ctx.solve(mode=hs.MODE_CLASSIC, alpha=3.0, max_iter=4, term_type=ITER)
        ctx.solve(mode=hs.MODE_CLASSIC, alpha=3.0, max_iter=5, term_type=ITER, use_previous=True, reuse_derivatives=True)
        u, v = ctx.flow()
        uo, vo = oracle.classic_flow(A, B, 3.0, 9)
        assert np.array_equal(u, uo) and np.array_equal(v, vo)
        ctx.solve(lam=0.5, max_iter=6, term_type=ITER)
        u, v = ctx.flow()
        uo, vo = oracle.calc_optical_flow_hs(A, B, 0.5, 6, term_type=ITER)
        assert rms(u, uo) <= 1e-4 and rms(v, vo) <= 1e-4
        with pytest.raises(hs.HsflowError):
            ctx.solve(mode=hs.MODE_CLASSIC, alpha=1.0, max_iter=5, term_type=ITER | EPS)


def test_cpp_dropin_cli(hs, oracle, gpu_ok, tmp_path):
    """The reference's positional command line, served by the C++ class over the C ABI."""
    cli = os.path.join(ROOT, "opticalflowhs_amd", "hsflow_cli")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "opticalflowhs_amd", "csrc"), "-s", "host"])
    W, H = 200, 120
    A, B = synth.translating_pair(W, H, seed=9, dx=2.5, dy=-1.5)
    p1, p2 = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")
    write_pnm(p1, A)
    write_pnm(p2, B)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "opticalflowhs_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    # -cv route: blur + HS(lambda) + arrows for |flow| > 1
    out_cv = str(tmp_path / "cv.ppm")
    r = subprocess.run([cli, "-cv", "-hd", p1, p2, out_cv, "0.1", "60"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Avg time" in r.stdout, r.stdout + r.stderr
    img = read_ppm(out_cv)
    fa, fb = oracle.box_blur3(A), oracle.box_blur3(B)
    uo, vo = oracle.calc_optical_flow_hs(fa, fb, 0.1, 60, float(np.float32(1e-6)), ITER | EPS)
    want_dots = sum(1 for y in range(0, H, 4) for x in range(0, W, 4) if abs(uo[y, x]) > 1 or abs(vo[y, x]) > 1)
    dots = sum(1 for y in range(0, H, 4) for x in range(0, W, 4) if img[y, x].any())  # dot centre, possibly under the line
    assert want_dots > 20 and abs(dots - want_dots) <= max(2, want_dots // 50)
    # -cl route: alpha, fixed iteration count, GPU
    out_cl = str(tmp_path / "cl.ppm")
    r = subprocess.run([cli, "-cl", "-hd", p1, p2, out_cl, "3", "40", "1", "GPU"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and os.path.getsize(out_cl) == W * H * 3 + len(b"P6\n%d %d\n255\n" % (W, H))
    # refused / malformed invocations keep the reference's behaviour
    r = subprocess.run([cli, "-cl", "-hd", p1, p2, out_cl, "3", "40", "1", "CPU"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "CPU" in r.stdout
    r = subprocess.run([cli, "-cv", "-hd", p1, p2], env=env, capture_output=True, text=True, timeout=60)
    assert "Wrong argument list" in r.stdout
    r = subprocess.run([cli, "-cv", "-hd", str(tmp_path / "missing.pgm"), p2, out_cv, "0.1", "5"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and "Input image error" in r.stdout


def _cli(args, tmp_path, extra_env=None):
    cli = os.path.join(ROOT, "opticalflowhs_amd", "hsflow_cli")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "opticalflowhs_amd", "csrc"), "-s", "host"])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "opticalflowhs_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    env.update(extra_env or {})
    r = subprocess.run([cli] + args, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return r


@pytest.mark.parametrize("name", ["city", "bunny"])
def test_hip_path_reproduces_the_cpu_route_pictures(hs, oracle, gpu_ok, tmp_path, name):
    """The pictures the reference's CPU route wrote (cvCalcOpticalFlowHS inside; tests/refpics.py) vs
    the HIP path: through the Python mirror (blur + ITER|EPS solve on the device) and through the C++
    drop-in's `-cv` command line.  Saved as JPEG, both drawings must BE the reference's picture."""
    pytest.importorskip("PIL")
    import refpics
    A0, B0 = refpics.gray_pair(name)
    H, W = A0.shape
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames_gray_blur(A0, B0)
        info = ctx.solve(lam=refpics.LAMBDA, max_iter=refpics.ITERATIONS, epsilon=refpics.EPSILON, term_type=ITER | EPS)
        u, v = ctx.flow()
    assert info["iterations_done"] == refpics.ITERATIONS
    wrong, quality = refpics.picture_difference(refpics.render(u, v), name)
    assert wrong == 0, (wrong, quality)
    uo, vo = oracle.calc_optical_flow_hs(oracle.box_blur3(A0), oracle.box_blur3(B0), refpics.LAMBDA, refpics.ITERATIONS,
                                         refpics.EPSILON, ITER | EPS)
    assert rms(u, uo) <= 1e-4 and rms(v, vo) <= 1e-4
    out = str(tmp_path / "out.ppm")
    _cli(["-cv", "-hd", os.path.join(GOLDEN, name + "_1_gray.pgm"), os.path.join(GOLDEN, name + "_2_gray.pgm"), out, ".1", "10"], tmp_path)
    drawn = read_ppm(out)
    assert np.array_equal(drawn, refpics.render(u, v))          # the C++ drawing = the rule as restated in refpics
    assert refpics.picture_difference(drawn, name)[0] == 0


@pytest.mark.parametrize("name", ["city", "bunny"])
def test_hip_path_reproduces_the_opencl_route_pictures(hs, oracle, gpu_ok, tmp_path, name):
    """The pictures the reference's OpenCL route wrote (Kernels.cl as shipped: v never written) vs the
    HIP classic kernels in MODE_CLASSIC_AS_SHIPPED, via the Python mirror and via `-cl` of the drop-in."""
    pytest.importorskip("PIL")
    import refpics
    A, B = refpics.gray_pair(name)
    H, W = A.shape
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ctx.solve(mode=hs.MODE_CLASSIC_AS_SHIPPED, alpha=refpics.ALPHA, max_iter=refpics.ITERATIONS, term_type=ITER)
        u, v = ctx.flow()
        uo, vo = oracle.classic_flow(A, B, refpics.ALPHA, refpics.ITERATIONS, update_v=False)
        assert np.array_equal(u, uo) and not v.any()
        ctx.solve(mode=hs.MODE_CLASSIC, alpha=refpics.ALPHA, max_iter=refpics.ITERATIONS, term_type=ITER)  # intended scheme
        u2, v2 = ctx.flow()
        uo2, vo2 = oracle.classic_flow(A, B, refpics.ALPHA, refpics.ITERATIONS)
        assert np.array_equal(u2, uo2) and np.array_equal(v2, vo2) and v2.any()
    assert refpics.picture_difference(refpics.render(u, v, "cl"), name, "cl")[0] == 0
    out = str(tmp_path / "out.ppm")
    args = ["-cl", "-hd", os.path.join(GOLDEN, name + "_1_gray.pgm"), os.path.join(GOLDEN, name + "_2_gray.pgm"), out, "15", "10", "1", "GPU"]
    _cli(args, tmp_path, {"HSFLOW_CL_AS_SHIPPED": "1"})
    drawn = read_ppm(out)
    assert np.array_equal(drawn, refpics.render(u, v, "cl"))
    assert refpics.picture_difference(drawn, name, "cl")[0] == 0
    _cli(args, tmp_path)                                        # default: v update restored -> a different picture
    assert refpics.picture_difference(read_ppm(out), name, "cl")[0] > 200


def test_classic_fused_kernel_equals_single_sweep_kernel(hs, oracle, gpu_ok):
    """The multi-sweep LDS-tile kernel of the classic mode against the one-sweep kernel and the oracle,
    bit for bit: tile / depth / workgroup variants, ragged sizes, several pairs, the as-shipped form."""
    rng = np.random.default_rng(7)
    for case, (W, H) in enumerate([(200, 120), (37, 29), (5, 3), (1, 9), (9, 1), (131, 67), (258, 64)]):
        A, B = synth.translating_pair(W, H, seed=60 + case, dx=1.5, dy=-0.5) if min(W, H) > 8 else synth.random_pair(W, H, seed=60 + case)
        it = int(rng.integers(1, 30))
        alpha = float(rng.uniform(0.5, 20.0))
        uo, vo = oracle.classic_flow(A, B, alpha, it)
        us, vs = oracle.classic_flow(A, B, alpha, it, update_v=False)
        with hs.HSFlow(W, H, own_stream=True) as ctx:
            ctx.set_frames(A, B)
            variants = [dict(kernel=hs.KERNEL_SIMPLE), dict(), dict(kernel=hs.KERNEL_FUSED, fuse_steps=1),
                        dict(kernel=hs.KERNEL_FUSED, fuse_steps=int(rng.integers(2, 13))),
                        dict(kernel=hs.KERNEL_FUSED, fuse_steps=4, tile_w=32, tile_h=8, threads=256),
                        dict(kernel=hs.KERNEL_FUSED, fuse_steps=3, threads=512),
                        dict(kernel=hs.KERNEL_FUSED, fuse_steps=6, threads=1024)]
            for kw in variants:
                info = ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=it, term_type=ITER, **kw)
                u, v = ctx.flow()
                assert info["iterations_done"] == it
                assert np.array_equal(u, uo) and np.array_equal(v, vo), (W, H, it, alpha, kw, info)
            info = ctx.solve(mode=hs.MODE_CLASSIC_AS_SHIPPED, alpha=alpha, max_iter=it, term_type=ITER)
            u, v = ctx.flow()
            assert info["kernel"] in (hs.KERNEL_FUSED, hs.KERNEL_STRIP) and np.array_equal(u, us) and not v.any()
            with pytest.raises(hs.HsflowError):
                ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=it, term_type=ITER, kernel=hs.KERNEL_FOLD)
    # a batch of pairs in one context
    W, H, n = 96, 40, 3
    with hs.HSFlow(W, H, n, own_stream=True) as ctx:
        pairs = [synth.translating_pair(W, H, seed=80 + i) for i in range(n)]
        for i, (A, B) in enumerate(pairs):
            ctx.set_frames(A, B, pair=i)
        ctx.solve(mode=hs.MODE_CLASSIC, alpha=4.0, max_iter=11, term_type=ITER)
        for i, (A, B) in enumerate(pairs):
            u, v = ctx.flow(pair=i)
            uo, vo = oracle.classic_flow(A, B, 4.0, 11)
            assert np.array_equal(u, uo) and np.array_equal(v, vo), i


def test_classic_strip_kernel_bit_exact(hs, oracle, gpu_ok):
    """The register-strip kernel of the classic mode (k_classic_strip) against oracle/hs_classic_oracle.c, bit for bit:
    every compiled row count, several tiles in both directions, widths that are no multiple of 4, heights no row count
    divides (the last tile row is aligned to the bottom border), launches that split the sweeps, flat areas beside
    moving texture (flow values down to denormals reach the division), warm start, the as-shipped form, several pairs."""
    rng = np.random.default_rng(11)
    refused = 0
    cases = [(600, 300), (333, 131), (258, 64), (1000, 97), (64, 480), (37, 29), (517, 203), (1920, 200)]
    for case, (W, H) in enumerate(cases):
        A, B = synth.translating_pair(W, H, seed=160 + case, dx=1.5, dy=-0.5)
        if case % 2 == 0:  # a static flat area and a static textured one: t = 0 exactly and t -> tiny values
            B[: H // 3] = A[: H // 3]
            A[:, : W // 4] = 90
            B[:, : W // 4] = 90
        it = int(rng.integers(3, 40))
        alpha = float(rng.uniform(0.5, 20.0))
        uo, vo = oracle.classic_flow(A, B, alpha, it)
        with hs.HSFlow(W, H, own_stream=True) as ctx:
            ctx.set_frames(A, B)
            variants = [dict()] + [dict(strip_rows=r) for r in range(2, 9)] + \
                       [dict(fuse_steps=int(rng.integers(1, 17))), dict(strip_rows=4, threads=256, fuse_steps=3),
                        dict(strip_rows=8, threads=512, fuse_steps=12), dict(strip_rows=3, threads=1024, fuse_steps=7)]
            for kw in variants:
                try:
                    info = ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=it, term_type=ITER, kernel=hs.KERNEL_STRIP, **kw)
                except hs.HsflowError:  # no aligned shape for this height with the requested rows / threads
                    assert kw, (W, H)
                    refused += 1
                    continue
                u, v = ctx.flow()
                assert info["kernel"] == hs.KERNEL_STRIP and info["iterations_done"] == it
                assert np.array_equal(u, uo) and np.array_equal(v, vo), (W, H, it, alpha, kw, info,
                                                                         np.argwhere(u != uo)[:4], np.argwhere(v != vo)[:4])
            # warm start across two solves, then the as-shipped form (v never written)
            ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=2, term_type=ITER, kernel=hs.KERNEL_STRIP)
            ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=it - 2, term_type=ITER, kernel=hs.KERNEL_STRIP, use_previous=True,
                      reuse_derivatives=True)
            u, v = ctx.flow()
            assert np.array_equal(u, uo) and np.array_equal(v, vo), (W, H)
            ctx.solve(mode=hs.MODE_CLASSIC_AS_SHIPPED, alpha=alpha, max_iter=it, term_type=ITER, kernel=hs.KERNEL_STRIP)
            u, v = ctx.flow()
            us, _ = oracle.classic_flow(A, B, alpha, it, update_v=False)
            assert np.array_equal(u, us) and not v.any()
    assert refused < 4 * len(cases)
    # the division: flow that decays into a static textured area runs through every magnitude down to the denormals (a small
    # alpha makes it fall by orders of magnitude per pixel), so the numerator meets v_div_scale's rescaling and 0 / den
    W, H = 320, 96
    A, B = synth.translating_pair(W, H, seed=177, dx=1.5, dy=-0.5)
    B[:, : W // 2] = A[:, : W // 2]
    for alpha, it in ((0.02, 60), (0.3, 60)):
        uo, vo = oracle.classic_flow(A, B, alpha, it)
        tiny = np.abs(uo[uo != 0])
        assert tiny.min() < 1e-36 and (uo == 0).any() and tiny.max() > 1e-3, (tiny.min(), tiny.max())
        with hs.HSFlow(W, H, own_stream=True) as ctx:
            ctx.set_frames(A, B)
            for kw in (dict(), dict(strip_rows=6), dict(strip_rows=4), dict(strip_rows=2, fuse_steps=3),
                       dict(strip_rows=5), dict(strip_rows=8, fuse_steps=5), dict(strip_rows=7)):  # (5, 7, 8: the complete division, packed)
                info = ctx.solve(mode=hs.MODE_CLASSIC, alpha=alpha, max_iter=it, term_type=ITER, kernel=hs.KERNEL_STRIP, **kw)
                u, v = ctx.flow()
                assert np.array_equal(u, uo) and np.array_equal(v, vo), (alpha, kw, info, np.argwhere(u != uo)[:4])
            with pytest.raises(hs.HsflowError):  # far outside the range the precomputed reciprocal is proven for
                ctx.solve(mode=hs.MODE_CLASSIC, alpha=1e-9, max_iter=3, term_type=ITER, kernel=hs.KERNEL_STRIP)
            ctx.solve(mode=hs.MODE_CLASSIC, alpha=1e-9, max_iter=3, term_type=ITER)  # AUTO: the LDS-tile kernel takes it
            u, v = ctx.flow()
            uo9, vo9 = oracle.classic_flow(A, B, 1e-9, 3)
            assert ctx.info()["kernel"] == hs.KERNEL_FUSED and np.array_equal(u, uo9, equal_nan=True) and np.array_equal(v, vo9, equal_nan=True)
    W, H, n = 300, 160, 3
    with hs.HSFlow(W, H, n, own_stream=True) as ctx:
        pairs = [synth.translating_pair(W, H, seed=190 + i) for i in range(n)]
        for i, (A, B) in enumerate(pairs):
            ctx.set_frames(A, B, pair=i)
        ctx.solve(mode=hs.MODE_CLASSIC, alpha=4.0, max_iter=13, term_type=ITER, kernel=hs.KERNEL_STRIP)
        for i, (A, B) in enumerate(pairs):
            u, v = ctx.flow(pair=i)
            uo, vo = oracle.classic_flow(A, B, 4.0, 13)
            assert np.array_equal(u, uo) and np.array_equal(v, vo), i


@pytest.mark.parametrize("name", ["city", "bunny"])
def test_cli_on_the_reference_jpegs_reproduces_its_pictures(hs, gpu_ok, tmp_path, name):
    """The reference's command lines on the reference's own JPEG files (main.cpp:16,20 defaults with the
    10 iterations its pictures were made with): what the drop-in writes, saved as JPEG, IS the reference's
    picture -- for the CPU route, and for the OpenCL route with Kernels.cl as shipped -- and, written as
    .jpg, the reference's output file itself."""
    pytest.importorskip("PIL")
    import refpics
    a, b = os.path.join(GOLDEN, "ref_%s_1.jpg" % name), os.path.join(GOLDEN, "ref_%s_2.jpg" % name)
    out = str(tmp_path / "out.ppm")
    _cli(["-cv", "-hd", a, b, out, ".1", "10"], tmp_path)
    assert refpics.picture_difference(read_ppm(out), name, "cv")[0] == 0
    _cli(["-cl", "-hd", a, b, out, "15", "10", "1", "GPU"], tmp_path, {"HSFLOW_CL_AS_SHIPPED": "1"})
    assert refpics.picture_difference(read_ppm(out), name, "cl")[0] == 0
    # ... and with a .jpg output name, like the reference's own runs: the very same FILE, byte for byte
    out = str(tmp_path / "out.jpg")
    _cli(["-cv", "-hd", a, b, out, ".1", "10"], tmp_path)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "ref_%s_cv_out.jpg" % name), "rb").read()
    _cli(["-cl", "-hd", a, b, out, "15", "10", "1", "GPU"], tmp_path, {"HSFLOW_CL_AS_SHIPPED": "1"})
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "ref_%s_cl_out.jpg" % name), "rb").read()
    if name == "bunny":
        # the reference's fifth picture (Release/bunny_cl_out.jpg): the same command line with TWO iterations
        # (tools/scan_release_bunny.py found them) -- again the very same file
        _cli(["-cl", "-hd", a, b, out, "15", "2", "1", "GPU"], tmp_path, {"HSFLOW_CL_AS_SHIPPED": "1"})
        assert open(out, "rb").read() == open(os.path.join(GOLDEN, "ref_bunny_cl_out_release.jpg"), "rb").read()


def test_cv_camera_route_matches_the_reference_loop(hs, oracle, gpu_ok, tmp_path):
    """`-cv -cam` on numbered frame files: the reference's loop (OpticalFlowOpenCV.cpp:56-131) blurs in place
    and re-uses the blurred new frame as the next old one, so from the second pair on the old frame is blurred
    twice.  The drawings must be those of the oracle run through exactly that loop."""
    import refpics
    W, H, n = 160, 96, 4
    rng = np.random.default_rng(3)
    cam, out = tmp_path / "cam", tmp_path / "out"
    cam.mkdir()
    out.mkdir()
    frames = []
    for i in range(n):
        A, _ = synth.translating_pair(W, H, seed=77, dx=1.5 * i, dy=-0.75 * i)
        frames.append(A)
        write_pnm(str(cam / ("frame_%04d.pgm" % i)), A)
    r = _cli(["-cv", "-cam", ".1", "12"], tmp_path, {"HSFLOW_CAMERA_DIR": str(cam), "HSFLOW_CAMERA_OUT": str(out)})
    assert "Avg time" in r.stdout
    old = frames[0]
    for i in range(1, n):
        old_b, new_b = oracle.box_blur3(old), oracle.box_blur3(frames[i])
        u, v = oracle.calc_optical_flow_hs(old_b, new_b, 0.1, 12, float(np.float32(1e-6)), ITER | EPS)
        assert np.array_equal(read_ppm(str(out / ("flow_%04d.ppm" % i))), refpics.render(u, v)), i
        old = new_b                                   # already blurred; blurred again next time round
    assert not (out / ("flow_%04d.ppm" % n)).exists()
