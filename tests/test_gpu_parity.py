"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bar (BASELINE.json north_star): u and v each within 1e-4 RMS of the oracle on identical u8
inputs.  The oracle evaluates in double with fp32 stores (x87 semantics of cv210.dll), the GPU in
pure fp32, so bit equality is not expected between the two; it IS expected between any two GPU
kernel variants (simple / fused, any tile, any fuse depth, graph or not), which is asserted.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from opticalflowhs_amd import synth

pytestmark = pytest.mark.gpu

ITER, EPS = 1, 2
RMS_TOL = 1e-4          # the stated tolerance
_report = {}


def rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def check(name, got, want, tol=RMS_TOL):
    (u, v), (uo, vo) = got, want
    ru, rv = rms(u, uo), rms(v, vo)
    mx = float(max(np.abs(u.astype(np.float64) - uo).max(), np.abs(v.astype(np.float64) - vo).max()))
    _report[name] = {"rms_u": ru, "rms_v": rv, "max_abs": mx, "flow_max": float(max(np.abs(uo).max(), np.abs(vo).max()))}
    assert np.isfinite(u).all() and np.isfinite(v).all(), name
    assert ru <= tol and rv <= tol, (name, ru, rv, mx)


@pytest.fixture(scope="module", autouse=True)
def write_report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(_report, f, indent=1, sort_keys=True)
    except OSError:
        pass


def gpu_solve(hs, A, B, lam, it, eps=1e-6, tt=ITER, **kw):
    H, W = A.shape
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        info = ctx.solve(lam=lam, max_iter=it, epsilon=eps, term_type=tt, **kw)
        u, v = ctx.flow()
    return u, v, info


def test_native_library_is_loaded(hs, gpu_ok):
    assert os.path.samefile(hs._lib.load()._name, os.path.join(ROOT, "opticalflowhs_amd", "libhsflow.so"))
    maps = open("/proc/self/maps").read()
    assert "libhsflow.so" in maps


@pytest.mark.parametrize("shape", [(29, 37), (48, 64), (240, 424), (1, 9), (9, 1), (2, 2), (3, 3), (1, 1), (5, 260), (130, 5)])
def test_derivatives_bit_exact(hs, oracle, gpu_ok, shape):
    H, W = shape
    A, B = synth.random_pair(W, H, seed=H * 31 + W)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ctx.solve(lam=1.0, max_iter=1, term_type=ITER)
        dx, dy, dt = ctx.derivatives()
        a, b = ctx.frames()
    Ix, Iy, It = oracle.derivatives(A, B)
    assert np.array_equal(a, A) and np.array_equal(b, B)
    assert np.array_equal(dx, Ix) and np.array_equal(dy, Iy) and np.array_equal(dt, It)


@pytest.mark.parametrize("shape", [(29, 37), (48, 64), (1, 9), (9, 1), (2, 2), (3, 3), (1, 1), (7, 130), (131, 6)])
@pytest.mark.parametrize("kernel", ["simple", "fused", "strip", "fold"])
def test_small_random_pairs(hs, oracle, gpu_ok, shape, kernel):
    H, W = shape
    A, B = synth.random_pair(W, H, seed=H * 131 + W)
    k = {"simple": hs.KERNEL_SIMPLE, "fused": hs.KERNEL_FUSED, "strip": hs.KERNEL_STRIP, "fold": hs.KERNEL_FOLD}[kernel]
    for lam in (0.01, 1.0, 10.0):
        for it in (1, 2, 10, 100):
            uo, vo = oracle.calc_optical_flow_hs(A, B, lam, it, term_type=ITER)
            u, v, _ = gpu_solve(hs, A, B, lam, it, kernel=k)
            check("rand_%dx%d_%s_l%g_i%d" % (W, H, kernel, lam, it), (u, v), (uo, vo))


def test_golden_k5(hs, gpu_ok):
    d = np.load(os.path.join(GOLDEN, "k5_random.npz"))
    for (W, H) in ((37, 29), (64, 48)):
        A, B = d["A_%dx%d" % (W, H)], d["B_%dx%d" % (W, H)]
        for lam in (0.01, 0.1, 1.0, 10.0):
            for it in (1, 2, 10, 100):
                u, v, _ = gpu_solve(hs, A, B, lam, it)
                check("k5_%dx%d_l%g_i%d" % (W, H, lam, it), (u, v),
                      (d["u_%dx%d_l%g_i%d" % (W, H, lam, it)], d["v_%dx%d_l%g_i%d" % (W, H, lam, it)]))


def test_golden_synth_and_bunny(hs, oracle, gpu_ok):
    d = np.load(os.path.join(GOLDEN, "synth_256x128_s1_l1_i100.npz"))
    u, v, _ = gpu_solve(hs, d["A"], d["B"], 1.0, 100)
    check("synth_256x128", (u, v), (d["u"], d["v"]))
    from PIL import Image  # PGM decoding only
    fr = [oracle.box_blur3(np.asarray(Image.open(os.path.join(GOLDEN, "bunny_%d_gray.pgm" % i)))) for i in (1, 2)]
    b = np.load(os.path.join(GOLDEN, "bunny_flow_l1_i50.npz"))
    # BASELINE config C1: lambda 1, 50 iterations, ITER|EPS with eps 1e-6 as the reference calls it
    u, v, info = gpu_solve(hs, fr[0], fr[1], 1.0, 50, eps=float(np.float32(1e-6)), tt=ITER | EPS)
    assert info["iterations_done"] == int(b["iters"]) == 50
    check("bunny_424x240_l1_i50", (u, v), (b["u"], b["v"]))


def test_kernel_variants_are_bit_identical(hs, gpu_ok):
    A, B = synth.smooth_random_pair(203, 117, seed=4, shift=(1, 1))
    ref = None
    variants = [dict(kernel=hs.KERNEL_SIMPLE)]
    for T in (1, 2, 3, 5, 8, 12):
        variants.append(dict(kernel=hs.KERNEL_FUSED, fuse_steps=T))
    variants += [dict(kernel=hs.KERNEL_FUSED, fuse_steps=4, tile_w=32, tile_h=16, threads=256),
                 dict(kernel=hs.KERNEL_FUSED, fuse_steps=4, tile_w=64, tile_h=24, threads=512),
                 dict(kernel=hs.KERNEL_FUSED, fuse_steps=6, tile_w=96, tile_h=40, threads=1024),
                 dict(kernel=hs.KERNEL_FUSED, fuse_steps=7, tile_w=204, tile_h=7),
                 dict(kernel=hs.KERNEL_FUSED, fuse_steps=5, use_graph=True),
                 dict(kernel=hs.KERNEL_STRIP), dict(kernel=hs.KERNEL_STRIP, use_graph=True)]
    for T, R, nt in ((1, 1, 256), (2, 2, 192), (3, 3, 1024), (4, 4, 1024), (5, 5, 1024), (7, 6, 768), (9, 7, 512),
                     (12, 8, 512), (8, 3, 640), (30, 8, 512), (6, 1, 1024)):
        variants.append(dict(kernel=hs.KERNEL_STRIP, fuse_steps=T, strip_rows=R, threads=nt))
    variants += [dict(kernel=hs.KERNEL_FOLD), dict(kernel=hs.KERNEL_FOLD, use_graph=True)]
    for T, R, nt in ((1, 1, 128), (2, 2, 128), (3, 3, 1024), (4, 4, 1024), (5, 5, 768), (7, 6, 512), (9, 7, 512),
                     (12, 8, 512), (8, 3, 320), (30, 8, 512), (6, 1, 1024), (11, 4, 448)):
        variants.append(dict(kernel=hs.KERNEL_FOLD, fuse_steps=T, strip_rows=R, threads=nt))
    variants += [
                 dict(kernel=hs.KERNEL_SIMPLE, use_graph=True)]
    for kw in variants:
        u, v, info = gpu_solve(hs, A, B, 0.2, 37, **kw)
        assert info["iterations_done"] == 37
        if ref is None:
            ref = (u, v)
        else:
            assert np.array_equal(u, ref[0]) and np.array_equal(v, ref[1]), kw


def test_derivative_pass_fused_into_first_launch(hs, oracle, gpu_ok):
    """The strip kernel's first launch computing the derivatives itself (k_jacobi_strip_deriv) against the
    separate derivative kernel (profile=True never fuses): the same flow bits, and the derivative plane it
    leaves behind equals the oracle's -- over image borders in every tile position, mirrored halos, all
    three Eps variants, warm starts and shapes where the fusion must be declined."""
    cases = [(1920, 1080), (256, 96), (260, 131), (512, 200), (1024, 333), (300, 257), (258, 140), (264, 80), (2048, 97),
             (424, 240), (600, 480), (128, 64), (132, 33), (124, 300), (854, 480), (1366, 768), (261, 100), (259, 97),
             (130, 70), (257, 83)]
    for n, (W, H) in enumerate(cases):
        A, B = synth.random_pair(W, H, seed=70 + n)
        Ix, Iy, It = oracle.derivatives(A, B)
        with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
            ctx.set_frames(A, B)
            for it, R, graph, kern in ((1, 5, False, hs.KERNEL_STRIP), (7, 4, True, hs.KERNEL_STRIP), (25, 5, True, hs.KERNEL_STRIP),
                                       (43, 6, False, hs.KERNEL_STRIP), (20, 0, True, hs.KERNEL_STRIP), (9, 2, True, hs.KERNEL_STRIP),
                                       (5, 7, False, hs.KERNEL_STRIP), (10, 0, True, hs.KERNEL_AUTO),
                                       (1, 1, False, hs.KERNEL_FOLD), (13, 2, True, hs.KERNEL_FOLD), (8, 3, False, hs.KERNEL_FOLD),
                                       (30, 4, True, hs.KERNEL_FOLD), (17, 5, False, hs.KERNEL_FOLD), (6, 6, True, hs.KERNEL_FOLD),
                                       (50, 0, True, hs.KERNEL_FOLD)):
                kw = dict(lam=0.5, max_iter=it, kernel=kern, strip_rows=R)
                for tt in (ITER, ITER | EPS):
                    for prev in (False, True):
                        if prev:  # the warm start: the flow a short solve leaves on the device
                            ctx.solve(lam=2.0, max_iter=3, term_type=ITER, kernel=hs.KERNEL_SIMPLE)
                        i1 = ctx.solve(term_type=tt, epsilon=1e-6, use_previous=prev, use_graph=graph, **kw)
                        u1, v1 = ctx.flow()
                        dx, dy, dt = ctx.derivatives()
                        fold = i1["kernel"] == hs.KERNEL_FOLD
                        rows = (i1["threads"] // 64) * i1["groups_per_thread"] * (2 if fold else 1)
                        fusable = i1["kernel"] in (hs.KERNEL_STRIP, hs.KERNEL_FOLD) and \
                            W >= (128 if fold else 256) and H >= rows and i1["groups_per_thread"] <= 6
                        if tt == ITER:
                            assert i1["deriv_fused"] == (1 if fusable else 0), (W, H, it, R, tt, prev, i1)
                        else:  # ITER|EPS: plans whose strips have no core row at an edge run the exact pass (separate kernel)
                            assert i1["deriv_fused"] in ((0, 1) if fusable else (0,)), (W, H, it, R, tt, prev, i1)
                        assert np.array_equal(dx, Ix) and np.array_equal(dy, Iy) and np.array_equal(dt, It), (W, H, it, R, tt, prev)
                        if prev:
                            ctx.solve(lam=2.0, max_iter=3, term_type=ITER, kernel=hs.KERNEL_SIMPLE)
                        i2 = ctx.solve(term_type=tt, epsilon=1e-6, use_previous=prev, profile=True, **kw)
                        u2, v2 = ctx.flow()
                        assert i2["deriv_fused"] == 0
                        assert i1["iterations_done"] == i2["iterations_done"] == it
                        assert np.array_equal(u1, u2) and np.array_equal(v1, v2), (W, H, it, R, tt, prev)
            # a fresh pair on the same context: the plane must be recomputed, not reused
            A2, B2 = synth.random_pair(W, H, seed=170 + n)
            ctx.set_frames(A2, B2)
            ctx.solve(lam=0.5, max_iter=9, term_type=ITER, kernel=hs.KERNEL_AUTO, use_graph=True)
            u1, v1 = ctx.flow()
            dx, dy, dt = ctx.derivatives()
            J = oracle.derivatives(A2, B2)
            assert np.array_equal(dx, J[0]) and np.array_equal(dy, J[1]) and np.array_equal(dt, J[2])
            ctx.solve(lam=0.5, max_iter=9, term_type=ITER, kernel=hs.KERNEL_STRIP, profile=True)
            u2, v2 = ctx.flow()
            assert np.array_equal(u1, u2) and np.array_equal(v1, v2)


def test_scaled_state_matches_canonical_arithmetic_down_to_denormals(hs, gpu_ok):
    """Inside a launch the strip / folded kernels carry 4^k * u (no multiplication by 0.25 per sweep).  Powers
    of 4 commute with rounding, so they agree bit for bit with the single-sweep kernel -- except next to the
    denormal range (flow decaying geometrically into a flat region), where the single-sweep kernel's averages
    lose bits to gradual underflow and the scaled ones do not: there the two differ by a few denormal ulps
    (1.4e-45).  A small textured patch in an otherwise flat frame makes that happen."""
    W, H = 640, 400
    rng = np.random.default_rng(3)
    A = np.full((H, W), 90, np.uint8)
    A[40:90, 50:110] = rng.integers(0, 256, (50, 60), dtype=np.uint8)
    B = np.roll(A, 1, axis=1)
    tiny = float(np.finfo(np.float32).tiny)
    for it in (60, 150):
        ref = gpu_solve(hs, A, B, 1.0, it, kernel=hs.KERNEL_SIMPLE)
        for kern, kw in ((hs.KERNEL_STRIP, {}), (hs.KERNEL_FOLD, {}), (hs.KERNEL_STRIP, dict(fuse_steps=24)), (hs.KERNEL_FOLD, dict(fuse_steps=7))):
            got = gpu_solve(hs, A, B, 1.0, it, kernel=kern, **kw)
            for g, r in ((got[0], ref[0]), (got[1], ref[1])):
                safe = np.abs(r) >= 1e-30
                assert np.array_equal(g[safe], r[safe]), (it, kern, kw)
                assert np.all(np.abs(g[~safe].astype(np.float64) - r[~safe]) < 1e-41), (it, kern, kw)
        assert (np.abs(ref[0]) < tiny).any() and (np.abs(ref[0]) > 1e-3).any()   # both regimes are present
    # enormous lambda: the flow and the constant term grow like sqrt(lambda); the scaled state must not overflow
    # where the canonical arithmetic does not (launches are kept short beyond lambda = 1e20)
    A, B = synth.random_pair(300, 200, seed=8)
    for lam in (1e12, 1e19, 1e21, 1e30, 3e38):
        ref = gpu_solve(hs, A, B, lam, 70, kernel=hs.KERNEL_SIMPLE)
        for kern, kw in ((hs.KERNEL_STRIP, {}), (hs.KERNEL_FOLD, {}), (hs.KERNEL_STRIP, dict(fuse_steps=32)), (hs.KERNEL_FOLD, dict(fuse_steps=24))):
            got = gpu_solve(hs, A, B, lam, 70, kernel=kern, **kw)
            for g, r in ((got[0], ref[0]), (got[1], ref[1])):
                fin = np.isfinite(r)
                assert np.array_equal(np.isfinite(g), fin), (lam, kern, kw)
                assert np.array_equal(g[fin], r[fin]), (lam, kern, kw, float(np.abs(r[fin]).max()))


def check_stop(name, oracle, A, B, lam, got, n_gpu, want, n_oracle, u0=None, v0=None, done_before=0):
    """Flow of an EPS-terminated solve against the oracle.  The stopping sweep is integer work and must be the
    oracle's, except where a sweep's Eps lies within fp32 rounding of epsilon (the oracle measures its own x87
    iterates, the GPU its fp32 ones): then it may differ by ONE, and the flow is compared with the oracle run
    to the GPU's sweep count instead -- never skipped.  How often that happened goes into the parity report."""
    assert abs(n_gpu - n_oracle) <= 1, (name, n_gpu, n_oracle)
    _report.setdefault("_eps_stop_sweep", {"same": 0, "off_by_one": 0, "off_by_one_cases": []})
    if n_gpu == n_oracle:
        _report["_eps_stop_sweep"]["same"] += 1
        check(name, got, want)
        return
    _report["_eps_stop_sweep"]["off_by_one"] += 1
    _report["_eps_stop_sweep"]["off_by_one_cases"].append(name)
    kw = dict(use_previous=True, velx=u0, vely=v0) if u0 is not None else {}
    uo, vo = oracle.calc_optical_flow_hs(A, B, lam, n_gpu - done_before, term_type=ITER, **kw)
    check(name + "_at_gpu_count", got, (uo, vo))


def test_eps_termination_matches_oracle(hs, oracle, gpu_ok):
    d = np.load(os.path.join(GOLDEN, "eps_48x40_l0.002_e1e-3.npz"))
    for kw in (dict(kernel=hs.KERNEL_SIMPLE), dict(kernel=hs.KERNEL_FUSED), dict(kernel=hs.KERNEL_FUSED, fuse_steps=5),
               dict(kernel=hs.KERNEL_STRIP), dict(kernel=hs.KERNEL_STRIP, fuse_steps=7, strip_rows=2),
               dict(kernel=hs.KERNEL_FOLD), dict(kernel=hs.KERNEL_FOLD, fuse_steps=5, strip_rows=3)):
        u, v, info = gpu_solve(hs, d["A"], d["B"], 0.002, 500, eps=1e-3, tt=ITER | EPS, **kw)
        assert abs(info["iterations_done"] - int(d["iters"])) <= 1, info
        assert info["last_eps"] < 1e-3
        check_stop("eps_stop_%s_%d" % (kw["kernel"], kw.get("fuse_steps", 0)), oracle, d["A"], d["B"], 0.002, (u, v), info["iterations_done"],
                   (d["u"], d["v"]), int(d["iters"]))
    # identical frames: stops after the first sweep with zero flow (K1)
    A = d["A"]
    u, v, info = gpu_solve(hs, A, A, 0.1, 100, eps=float(np.float32(1e-6)), tt=ITER | EPS)
    assert info["iterations_done"] == 1 and not u.any() and not v.any()
    # EPS only
    u2, v2, info2 = gpu_solve(hs, d["A"], d["B"], 0.002, 0, eps=1e-3, tt=EPS)
    check_stop("eps_only_stop", oracle, d["A"], d["B"], 0.002, (u2, v2), info2["iterations_done"], (d["u"], d["v"]), int(d["iters"]))
    # ITER|EPS continuing from a previous flow: 5 sweeps, then the rest with the early stop
    with hs.HSFlow(48, 40, 1, own_stream=True) as ctx:
        ctx.set_frames(d["A"], d["B"])
        ctx.solve(lam=0.002, max_iter=5, term_type=ITER)
        info3 = ctx.solve(lam=0.002, max_iter=495, epsilon=1e-3, term_type=ITER | EPS, use_previous=True)
        u3, v3 = ctx.flow()
    check_stop("eps_stop_use_previous", oracle, d["A"], d["B"], 0.002, (u3, v3), info3["iterations_done"] + 5, (d["u"], d["v"]), int(d["iters"]))
    # budget reached before the threshold: plain ITER result
    u4, v4, info4 = gpu_solve(hs, d["A"], d["B"], 0.002, 40, eps=1e-3, tt=ITER | EPS)
    u5, v5, _ = gpu_solve(hs, d["A"], d["B"], 0.002, 40, tt=ITER)
    assert info4["iterations_done"] == 40 and np.array_equal(u4, u5) and np.array_equal(v4, v5)


def test_eps_only_gives_up_at_the_fp32_limit_cycle(hs, gpu_ok):
    """EPS termination without a budget and an epsilon the fp32 iteration can never reach: the original would
    spin forever; here the solve returns HSFLOW_E_NOTERM once Eps has stopped decreasing, with the flow kept."""
    d = np.load(os.path.join(GOLDEN, "eps_48x40_l0.002_e1e-3.npz"))
    with hs.HSFlow(48, 40, 1, own_stream=True) as ctx:
        ctx.set_frames(d["A"], d["B"])
        for kw in (dict(kernel=hs.KERNEL_AUTO), dict(kernel=hs.KERNEL_SIMPLE)):
            with pytest.raises(hs.HsflowError) as e:
                ctx.solve(lam=0.002, max_iter=0, epsilon=1e-30, term_type=EPS, **kw)
            assert e.value.status == hs._lib.E_NOTERM
            info = ctx.info()
            assert 4096 <= info["iterations_done"] < (1 << 20) and 0 <= info["last_eps"] < 1e-5, info
            u, v = ctx.flow()
            assert np.isfinite(u).all() and np.isfinite(v).all() and np.abs(u).max() > 1e-3
        # identical frames: Eps is exactly 0 after one sweep, any positive epsilon stops there; epsilon = 0 is refused
        ctx.set_frames(d["A"], d["A"])
        assert ctx.solve(lam=0.1, max_iter=0, epsilon=1e-30, term_type=EPS)["iterations_done"] == 1
        with pytest.raises(hs.HsflowError) as e:
            ctx.solve(lam=0.1, max_iter=0, epsilon=0.0, term_type=EPS)
        assert e.value.status == hs._lib.E_NOTERM


def test_use_previous_and_row_copies(hs, gpu_ok):
    import torch
    A, B = synth.random_pair(70, 33, seed=21)
    u10, v10, _ = gpu_solve(hs, A, B, 0.3, 10)
    with hs.HSFlow(70, 33, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        ctx.solve(lam=0.3, max_iter=4, term_type=ITER)
        ctx.solve(lam=0.3, max_iter=6, term_type=ITER, use_previous=True)
        u, v = ctx.flow()
        assert np.array_equal(u, u10) and np.array_equal(v, v10)
        # device row copies out / in
        ud = torch.empty((5, 70), dtype=torch.float32, device="cuda")
        vd = torch.empty((5, 70), dtype=torch.float32, device="cuda")
        ctx.flow_rows_to(ud, vd, 7, 5)
        ctx.synchronize()
        assert np.array_equal(ud.cpu().numpy(), u10[7:12]) and np.array_equal(vd.cpu().numpy(), v10[7:12])
        zu, ov = torch.zeros_like(ud), torch.ones_like(vd)
        torch.cuda.synchronize()  # the fills run on torch's stream, the copy on the context's
        ctx.set_flow_rows_from(zu, ov, 0, 5)
        ctx.synchronize()
        u2, v2 = ctx.flow()
        assert not u2[:5].any() and np.all(v2[:5] == 1) and np.array_equal(u2[5:], u10[5:])


def test_batch_of_pairs_and_device_frames(hs, oracle, gpu_ok):
    import torch
    W, H, N = 150, 90, 3
    pairs = [synth.translating_pair(W, H, seed=1000 + i) for i in range(N)]
    with hs.HSFlow(W, H, N, own_stream=True) as ctx:
        for i, (A, B) in enumerate(pairs):
            if i == 1:  # frames already on the device, with a row pitch
                ta = torch.zeros((H, W + 10), dtype=torch.uint8, device="cuda")
                tb = torch.zeros((H, W + 10), dtype=torch.uint8, device="cuda")
                ta[:, :W] = torch.from_numpy(A).cuda()
                tb[:, :W] = torch.from_numpy(B).cuda()
                torch.cuda.synchronize()
                ctx.set_frames(ta[:, :W], tb[:, :W], pair=i)
            else:
                ctx.set_frames(A, B, pair=i)
        ctx.solve(lam=1.0, max_iter=40, term_type=ITER)
        for i, (A, B) in enumerate(pairs):
            uo, vo = oracle.calc_optical_flow_hs(A, B, 1.0, 40, term_type=ITER, threads=0)
            check("batch%d" % i, ctx.flow(i), (uo, vo))
            u1, v1, _ = gpu_solve(hs, A, B, 1.0, 40)
            u, v = ctx.flow(i)
            assert np.array_equal(u, u1) and np.array_equal(v, v1)


def test_strided_host_buffers_and_one_shot(hs, oracle, gpu_ok):
    import ctypes
    W, H = 61, 23
    A, B = synth.random_pair(W, H, seed=5)
    big_a = np.zeros((H, 80), np.uint8)
    big_b = np.zeros((H, 80), np.uint8)
    big_a[:, :W], big_b[:, :W] = A, B
    velx = np.full((H, 72), 7.0, np.float32)
    vely = np.full((H, 72), 7.0, np.float32)
    L = hs._lib.load()
    st = L.hsflow_calc_optical_flow_hs_8u32f(big_a.ctypes.data, big_b.ctypes.data, 80, W, H, 0, velx.ctypes.data,
                                             vely.ctypes.data, 72 * 4, 0.5, ITER, 12, 0.0)
    assert st == 0, L.hsflow_last_error(None)
    uo, vo = oracle.calc_optical_flow_hs(A, B, 0.5, 12, term_type=ITER)
    check("one_shot", (velx[:, :W], vely[:, :W]), (uo, vo))
    assert np.all(velx[:, W:] == 7.0)  # padding untouched
    # python mirror of cvCalcOpticalFlowHS, with use_previous
    vx = np.zeros((H, W), np.float32)
    vy = np.zeros((H, W), np.float32)
    hs.calc_optical_flow_hs(A, B, 0, vx, vy, 0.5, hs.term_criteria(ITER, 5, 0))
    hs.calc_optical_flow_hs(A, B, 1, vx, vy, 0.5, hs.term_criteria(ITER, 7, 0))
    check("mirror_use_previous", (vx, vy), (uo, vo))
    # the one-shot keeps its context between calls: a stream of frames of one size, a size change, a
    # warm start and an error in between must all behave like independent calls
    import time
    frames = [synth.random_pair(W, H, seed=40 + i) for i in range(3)]
    t_first = t_later = 0.0
    for i, (Ai, Bi) in enumerate(frames * 2):
        vx2, vy2 = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        t0 = time.perf_counter()
        st = L.hsflow_calc_optical_flow_hs_8u32f(Ai.ctypes.data, Bi.ctypes.data, W, W, H, 0, vx2.ctypes.data, vy2.ctypes.data,
                                                 W * 4, 0.5, ITER | EPS, 9, 1e-6)
        dt = time.perf_counter() - t0
        assert st == 0
        ui, vi = oracle.calc_optical_flow_hs(Ai, Bi, 0.5, 9, 1e-6, ITER | EPS)
        check("one_shot_stream_%d" % i, (vx2, vy2), (ui, vi))
        if i > 0:
            t_later += dt / 5
    W2, H2 = 97, 31
    A2, B2 = synth.random_pair(W2, H2, seed=50)
    vx3, vy3 = np.zeros((H2, W2), np.float32), np.zeros((H2, W2), np.float32)
    assert L.hsflow_calc_optical_flow_hs_8u32f(A2.ctypes.data, B2.ctypes.data, W2, W2, H2, 0, vx3.ctypes.data, vy3.ctypes.data,
                                               W2 * 4, 2.0, ITER, 6, 0.0) == 0
    u4, v4 = oracle.calc_optical_flow_hs(A2, B2, 2.0, 4, term_type=ITER)
    vx4, vy4 = u4.copy(), v4.copy()                    # warm start from 4 sweeps, 2 more = 6 sweeps
    assert L.hsflow_calc_optical_flow_hs_8u32f(A2.ctypes.data, B2.ctypes.data, W2, W2, H2, 1, vx4.ctypes.data, vy4.ctypes.data,
                                               W2 * 4, 2.0, ITER, 2, 0.0) == 0
    assert np.array_equal(vx4, vx3) or rms(vx4, vx3) < 1e-5
    assert L.hsflow_calc_optical_flow_hs_8u32f(A2.ctypes.data, B2.ctypes.data, W2, W2, H2, 0, vx3.ctypes.data, vy3.ctypes.data,
                                               W2 * 4, -1.0, ITER, 6, 0.0) == hs._lib.E_ARG   # bad lambda drops the context
    assert L.hsflow_calc_optical_flow_hs_8u32f(A2.ctypes.data, B2.ctypes.data, W2, W2, H2, 0, vx3.ctypes.data, vy3.ctypes.data,
                                               W2 * 4, 2.0, ITER, 6, 0.0) == 0
    u6, v6 = oracle.calc_optical_flow_hs(A2, B2, 2.0, 6, term_type=ITER)
    check("one_shot_after_error", (vx3, vy3), (u6, v6))
    L.hsflow_release_cached()
    L.hsflow_release_cached()                          # idempotent


def test_error_statuses_on_gpu(hs, gpu_ok):
    A, B = synth.random_pair(16, 8, seed=1)
    with hs.HSFlow(16, 8, 1, own_stream=True) as ctx:
        with pytest.raises(hs.HsflowError) as e:
            ctx.solve(lam=1.0, max_iter=3, term_type=ITER)
        assert e.value.status == hs._lib.E_STATE
        ctx.set_frames(A, B)
        with pytest.raises(hs.HsflowError) as e:
            ctx.solve(lam=1.0, max_iter=0, term_type=ITER)
        assert e.value.status == hs._lib.E_NOTERM
        with pytest.raises(hs.HsflowError) as e:
            ctx.solve(lam=-1.0, max_iter=3, term_type=ITER)
        assert e.value.status == hs._lib.E_ARG
        with pytest.raises(hs.HsflowError):
            ctx.solve(lam=1.0, max_iter=3, term_type=0)
        with pytest.raises(hs.HsflowError) as e:
            ctx.solve(lam=1.0, max_iter=3, term_type=ITER, kernel=hs.KERNEL_FUSED, tile_w=4096, tile_h=4096)
        assert e.value.status == hs._lib.E_SIZE
        with pytest.raises(hs.HsflowError) as e:  # 2 rows x 1 wavefront cannot hold a 3-sweep halo
            ctx.solve(lam=1.0, max_iter=3, term_type=ITER, kernel=hs.KERNEL_STRIP, strip_rows=2, threads=64)
        assert e.value.status == hs._lib.E_SIZE
        with pytest.raises(ValueError):
            ctx.set_frames(A[:4], B[:4])
        with pytest.raises(hs.HsflowError):
            ctx.flow(pair=3)


def test_1080p_full_frame_parity(hs, oracle, gpu_ok):
    """BASELINE config C2: 1920x1080 translating texture (seed 1), lambda 1, 100 iterations."""
    A, B = synth.translating_pair(1920, 1080, seed=1)
    uo, vo = oracle.calc_optical_flow_hs(A, B, 1.0, 100, term_type=ITER, threads=0)
    u, v, info = gpu_solve(hs, A, B, 1.0, 100)
    assert info["kernel"] in (hs.KERNEL_STRIP, hs.KERNEL_FOLD)
    check("c2_1080p_auto", (u, v), (uo, vo))
    for k in (hs.KERNEL_FUSED, hs.KERNEL_FOLD, hs.KERNEL_STRIP):
        uf, vf, _ = gpu_solve(hs, A, B, 1.0, 100, kernel=k)
        assert np.array_equal(u, uf) and np.array_equal(v, vf), k
    us, vs, _ = gpu_solve(hs, A, B, 1.0, 100, kernel=hs.KERNEL_SIMPLE)
    assert np.array_equal(u, us) and np.array_equal(v, vs)
    ug, vg, _ = gpu_solve(hs, A, B, 1.0, 100, use_graph=True)
    assert np.array_equal(u, ug) and np.array_equal(v, vg)


def test_4k_full_frame_parity(hs, oracle, gpu_ok):
    """BASELINE config C3: 3840x2160 (seed 2), lambda 1, 200 iterations."""
    A, B = synth.translating_pair(3840, 2160, seed=2)
    uo, vo = oracle.calc_optical_flow_hs(A, B, 1.0, 200, term_type=ITER, threads=0)
    u, v, _ = gpu_solve(hs, A, B, 1.0, 200)
    check("c3_4k_auto", (u, v), (uo, vo))
    uf, vf, _ = gpu_solve(hs, A, B, 1.0, 200, kernel=hs.KERNEL_FOLD)
    assert np.array_equal(u, uf) and np.array_equal(v, vf)
    us, vs, _ = gpu_solve(hs, A, B, 1.0, 200, kernel=hs.KERNEL_SIMPLE)
    assert np.array_equal(u, us) and np.array_equal(v, vs)


def test_large_frame_size_independent_properties(hs, gpu_ok):
    """Row-slab-sized frame (a 16384-wide strip of config C5): properties that need no oracle.
    (1) identical frames give exactly zero flow; (2) T fused sweeps == T single sweeps;
    (3) flipping both frames left-right negates u and mirrors the flow up to rounding."""
    W, H = 16384, 512
    A, B = synth.translating_pair(W, H, seed=3)
    u, v, _ = gpu_solve(hs, A, A, 1.0, 20)
    assert not u.any() and not v.any()
    u8, v8, _ = gpu_solve(hs, A, B, 1.0, 24, kernel=hs.KERNEL_STRIP, fuse_steps=8)
    u1, v1, _ = gpu_solve(hs, A, B, 1.0, 24, kernel=hs.KERNEL_SIMPLE)
    assert np.array_equal(u8, u1) and np.array_equal(v8, v1)
    uf, vf, _ = gpu_solve(hs, np.ascontiguousarray(A[:, ::-1]), np.ascontiguousarray(B[:, ::-1]), 1.0, 24)
    assert rms(uf, -u8[:, ::-1]) < 1e-5 and rms(vf, v8[:, ::-1]) < 1e-5


def test_full_16384_frame_bands_against_oracle(hs, oracle, gpu_ok):
    """BASELINE config C5 at its full size on one GPU (268 Mpixel, 5.9 GB of planes): 8 sweeps, then
    three row bands (top border, middle, bottom border) are compared with the oracle run on the band
    plus a 16-row margin -- exact for the band because 8 sweeps cannot carry the margin's artificial
    border further than 8 rows."""
    W = H = 16384
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2), dtype=np.uint8)
    A = np.repeat(np.repeat(base, 8, axis=0), 8, axis=1)[3:3 + H, 5:5 + W].copy()   # blocky texture, cheap to make
    B = np.roll(A, (1, 2), axis=(0, 1))
    B[::7, ::5] += 3
    u, v, info = gpu_solve(hs, A, B, 1.0, 8)
    assert info["iterations_done"] == 8 and np.isfinite(u).all() and np.isfinite(v).all()
    for lo, hi in ((0, 40), (8000, 8040), (H - 40, H)):
        a, b = max(0, lo - 16), min(H, hi + 16)
        uo, vo = oracle.calc_optical_flow_hs(A[a:b], B[a:b], 1.0, 8, term_type=ITER, threads=0)
        check("c5_full_band_%d" % lo, (u[lo:hi], v[lo:hi]), (uo[lo - a:hi - a], vo[lo - a:hi - a]))


def test_c5_strip_500_sweeps_against_oracle_and_slab(hs, oracle, gpu_ok):
    """BASELINE config C5 at its own sweep count: the 16384 x 512 strip of the seed-3 frame that
    straddles the boundary between the first two of eight row slabs (frame rows 1792..2303, boundary at 2048),
    lambda 1, 500 sweeps, against the oracle's OpenMP form on the same strip (x87-vs-fp32 drift over 500 sweeps
    is what this pins: bar 1e-4 RMS); and the same strip through the row-slab driver with two ranks (threads,
    cut exactly at that boundary, halo 16: 31 exchanges) -- bit-identical to the one-context solve."""
    import threading
    from opticalflowhs_amd import slab
    from test_gpu_slab import LocalDist
    W, H, it = 16384, 512, 500
    A, B = synth.translating_pair(W, 16384, seed=3, row0=1792, rows=H)
    uo, vo = oracle.calc_optical_flow_hs(A, B, 1.0, it, term_type=ITER, threads=0)
    u, v, info = gpu_solve(hs, A, B, 1.0, it)
    assert info["iterations_done"] == it and info["kernel"] == hs.KERNEL_STRIP
    check("c5_strip_16384x512_i500", (u, v), (uo, vo))
    LocalDist.boxes = {}
    res, errs = {}, []

    def run(rank):
        try:
            s = slab.SlabSolver(LocalDist(rank), rank, 2, W, H, 16, lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, 0, first_row=r0))
            r0, r1 = s.local_frame_rows()
            s.set_frames(A[r0:r1], B[r0:r1])
            n_ex = s.solve(1.0, it)
            res[rank] = (s.lo, s.hi, n_ex) + s.owned_flow()
            s.close()
        except Exception as e:  # surfaced in the main thread
            errs.append((rank, repr(e)))

    ths = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in ths]
    [t.join(600) for t in ths]
    assert not errs and len(res) == 2, errs
    for rank in range(2):
        lo, hi, n_ex, ur, vr = res[rank]
        assert (lo, hi) == ((0, 256), (256, 512))[rank] and n_ex == 31
        assert np.array_equal(ur, u[lo:hi]) and np.array_equal(vr, v[lo:hi]), rank


def test_c4_shaped_batch_and_pipeline(hs, oracle, gpu_ok):
    """BASELINE config C4 at its per-GPU shape: 1920 x 1080 pairs with seeds 1000 + i, lambda 1, 100 sweeps --
    (1) eight of them resident in ONE context (one batch launch sequence), (2) the same pairs from page-locked
    host memory through the pair pipeline at depth 4 (uploads / downloads beside the solves).  Every pair must
    equal the single-pair context bit for bit; two sampled pairs are compared with the oracle."""
    W, H, N, it = 1920, 1080, 8, 100
    pairs = [synth.translating_pair(W, H, seed=1000 + i) for i in range(N)]
    single = []
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        for A, B in pairs:
            ctx.set_frames(A, B)
            ctx.solve(lam=1.0, max_iter=it, term_type=ITER, use_graph=True)
            single.append(ctx.flow())
    for i in (0, 5):
        uo, vo = oracle.calc_optical_flow_hs(pairs[i][0], pairs[i][1], 1.0, it, term_type=ITER, threads=0)
        check("c4_pair_%d" % i, single[i], (uo, vo))
    with hs.HSFlow(W, H, N, own_stream=True) as ctx:
        for i, (A, B) in enumerate(pairs):
            ctx.set_frames(A, B, pair=i)
        info = ctx.solve(lam=1.0, max_iter=it, term_type=ITER, use_graph=True)
        assert info["iterations_done"] == it and info["n_pairs"] == N
        for i in range(N):
            u, v = ctx.flow(i)
            assert np.array_equal(u, single[i][0]) and np.array_equal(v, single[i][1]), i
    with hs.PairPipeline(W, H, depth=4) as pl:
        bufs = []
        for A, B in pairs:
            a, b = hs.pinned_empty((H, W), np.uint8), hs.pinned_empty((H, W), np.uint8)
            a[...], b[...] = A, B
            u, v = hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)
            u.fill(np.nan)
            v.fill(np.nan)
            pl.submit(a, b, u, v, lam=1.0, max_iter=it, use_graph=True)
            bufs.append((u, v))
        pl.drain()
        for i, (u, v) in enumerate(bufs):
            assert np.array_equal(u, single[i][0]) and np.array_equal(v, single[i][1]), i
        # the same through the reference's own termination criteria (ITER|EPS, eps 1e-6: never fires here)
        eps6 = float(np.float32(1e-6))
        t = pl.submit(hs_pin(hs, pairs[3][0]), hs_pin(hs, pairs[3][1]), bufs[0][0], bufs[0][1], lam=1.0, max_iter=it, epsilon=eps6,
                      term_type=ITER | EPS, use_graph=True)
        i3 = pl.info(t)
        assert i3["iterations_done"] == it and i3["eps_rerun"] == 0
        assert np.array_equal(bufs[0][0], single[3][0]) and np.array_equal(bufs[0][1], single[3][1])


def hs_pin(hs, arr):
    out = hs.pinned_empty(arr.shape, arr.dtype)
    out[...] = arr
    return out


def test_randomized_shapes_parameters_kernels(hs, oracle, gpu_ok):
    """Seeded random sweep over frame shapes (including images narrower than the halo, odd widths,
    single rows / columns), lambda, sweep counts and every kernel with random tuning knobs: parity
    with the oracle and bit equality between kernels."""
    rng = np.random.default_rng(20261004)
    kernels = [hs.KERNEL_SIMPLE, hs.KERNEL_FUSED, hs.KERNEL_STRIP, hs.KERNEL_FOLD]
    for case in range(48):
        W = int(rng.integers(1, 340)) if case % 5 else int(rng.integers(1, 12))
        H = int(rng.integers(1, 260)) if case % 7 else int(rng.integers(1, 9))
        lam = float(10.0 ** rng.uniform(-3, 2))
        it = int(rng.integers(1, 64))
        if case % 3 == 0:
            A, B = synth.random_pair(W, H, seed=case)
        else:
            A, B = synth.translating_pair(W, H, seed=case, dx=float(rng.uniform(-2, 2)), dy=float(rng.uniform(-2, 2)))
        uo, vo = oracle.calc_optical_flow_hs(A, B, lam, it, term_type=ITER)
        ref = None
        for k in kernels:
            kw = dict(kernel=k)
            if k == hs.KERNEL_FUSED:
                kw["fuse_steps"] = int(rng.integers(1, 13))
            elif k in (hs.KERNEL_STRIP, hs.KERNEL_FOLD):
                T = int(rng.integers(1, 25))
                R = int(rng.integers(1, 9))
                fold = k == hs.KERNEL_FOLD
                maxw = (16 if R <= 4 else 12 if R == 5 else 8) if fold else (16 if R <= 5 else 12 if R == 6 else 8)
                need = -(-(2 * T + 1) // (R * (2 if fold else 1)))
                if need > maxw:
                    continue
                kw.update(fuse_steps=T, strip_rows=R, threads=64 * int(rng.integers(need, maxw + 1)))
            u, v, info = gpu_solve(hs, A, B, lam, it, **kw)
            assert info["iterations_done"] == it
            check("rnd%d_%dx%d_k%d" % (case, W, H, k), (u, v), (uo, vo))
            if ref is None:
                ref = (u, v)
            else:
                assert np.array_equal(u, ref[0]) and np.array_equal(v, ref[1]), (case, W, H, lam, it, kw)


def test_iter_eps_witness_and_fallback_regimes(hs, oracle, gpu_ok):
    """ITER|EPS on the strip kernel first runs "witness" launches that only PROVE Eps >= epsilon for
    all their sweeps, and falls back to measuring every sweep when the proof fails.  Both regimes
    must give the oracle's sweep count, Eps and flow."""
    W, H, it = 700, 300, 57
    A, B = synth.translating_pair(W, H, seed=21, dx=1.25, dy=-0.75)
    eps6 = float(np.float32(1e-6))
    uo, vo, n_o, e_o = oracle.calc_optical_flow_hs(A, B, 0.7, it, eps6, ITER | EPS, return_info=True)
    assert n_o == it                                   # natural images never get below 1e-6 this early
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        plain = ctx.solve(lam=0.7, max_iter=it, term_type=ITER)
        u0, v0 = ctx.flow()
        fast = ctx.solve(lam=0.7, max_iter=it, epsilon=eps6, term_type=ITER | EPS)
        u1, v1 = ctx.flow()
        assert fast["iterations_done"] == it and fast["jacobi_launches"] == plain["jacobi_launches"]  # proof held: no re-run
        assert np.array_equal(u0, u1) and np.array_equal(v0, v1)
        assert abs(fast["last_eps"] - e_o) <= 1e-3 * e_o
        check("iter_eps_witness", (u1, v1), (uo, vo))
        # last_eps: the strip kernel measures only the final sweep of its last launch (the others are witnessed),
        # the folded kernel every sweep of the last launch, the single-sweep kernel every sweep: the same number
        for k in (hs.KERNEL_FOLD, hs.KERNEL_SIMPLE, hs.KERNEL_FUSED):
            other = ctx.solve(lam=0.7, max_iter=it, epsilon=eps6, term_type=ITER | EPS, kernel=k)
            assert other["iterations_done"] == it and other["last_eps"] == fast["last_eps"], (k, other, fast)
        for T in (1, 2, 19, 57):   # last launch of 1 sweep (nothing witnessed), tails, a single launch
            r = ctx.solve(lam=0.7, max_iter=it, epsilon=eps6, term_type=ITER | EPS, kernel=hs.KERNEL_STRIP, fuse_steps=min(T, 24))
            assert r["iterations_done"] == it and r["eps_rerun"] == 0 and r["last_eps"] == fast["last_eps"], (T, r)
        # epsilon just under the final Eps: whether or not the sampled lower bounds still clear it (if
        # not, every sweep gets measured), the answer stays "budget reached"
        for frac in (0.98, 0.999, 0.99999):
            near = float(e_o) * frac
            uo2, vo2, n2, e2 = oracle.calc_optical_flow_hs(A, B, 0.7, it, near, ITER | EPS, return_info=True)
            slow = ctx.solve(lam=0.7, max_iter=it, epsilon=near, term_type=ITER | EPS)
            u2, v2 = ctx.flow()
            check_stop("iter_eps_near_%g" % frac, oracle, A, B, 0.7, (u2, v2), slow["iterations_done"], (uo2, vo2), n2)
        # the same pass replayed from a hipGraph (kernels, Eps reduction and read-back in one graph)
        for rep in range(3):
            g = ctx.solve(lam=0.7, max_iter=it, epsilon=eps6, term_type=ITER | EPS, use_graph=True)
            ug, vg = ctx.flow()
            assert g["iterations_done"] == it and g["eps_rerun"] == 0 and g["last_eps"] == fast["last_eps"]
            assert np.array_equal(ug, u1) and np.array_equal(vg, v1)
        ctx.solve(lam=0.7, max_iter=5, term_type=ITER)
        g2 = ctx.solve(lam=0.7, max_iter=it - 5, epsilon=eps6, term_type=ITER | EPS, use_graph=True, use_previous=True)
        ug, vg = ctx.flow()
        assert g2["iterations_done"] == it - 5 and np.array_equal(ug, u1) and np.array_equal(vg, v1)
        # epsilon above the final Eps: stops inside the budget, at the oracle's sweep
        over = float(e_o) * 3.0
        uo3, vo3, n3, e3 = oracle.calc_optical_flow_hs(A, B, 0.7, it, over, ITER | EPS, return_info=True)
        assert n3 < it
        stop = ctx.solve(lam=0.7, max_iter=it, epsilon=over, term_type=ITER | EPS, use_graph=True)
        u3, v3 = ctx.flow()
        assert stop["eps_rerun"] == 1
        assert stop["jacobi_launches"] > plain["jacobi_launches"]          # witness pass + exact pass + re-run
        check_stop("iter_eps_early_stop", oracle, A, B, 0.7, (u3, v3), stop["iterations_done"], (uo3, vo3), n3)


def test_cached_graphs_survive_growing_eps_buffers(hs, gpu_ok):
    """An ITER|EPS graph captured with small Eps buffers, then a solve that makes those buffers grow (more
    sweeps, more workgroups), then the first graph again: its replay must not use the freed buffers."""
    W, H = 640, 360
    A, B = synth.smooth_random_pair(W, H, seed=12, shift=(1, 0))
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        want = {}
        for it in (12, 700, 40):
            i = ctx.solve(lam=0.7, max_iter=it, term_type=ITER | EPS, epsilon=1e-7)
            want[it] = ctx.flow() + (i["iterations_done"], i["last_eps"])
        for rnd in range(3):
            for it in (12, 700, 12, 40, 700, 12):
                i = ctx.solve(lam=0.7, max_iter=it, term_type=ITER | EPS, epsilon=1e-7, use_graph=True)
                u, v = ctx.flow()
                assert np.array_equal(u, want[it][0]) and np.array_equal(v, want[it][1]), (rnd, it)
                assert (i["iterations_done"], i["last_eps"]) == want[it][2:], (rnd, it, i)


def test_graph_cache_stays_bounded(hs, oracle, gpu_ok):
    """A caller that changes lambda on every call creates a new captured sequence each time; the cache is
    trimmed instead of growing without bound, and results stay right across the trim."""
    W, H = 96, 64
    A, B = synth.translating_pair(W, H, seed=11)
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for i in range(70):
            lam = 0.5 + 0.01 * i
            ctx.solve(lam=lam, max_iter=6, term_type=ITER, use_graph=True)
            if i % 23 == 0 or i == 69:
                u, v = ctx.flow()
                uo, vo = oracle.calc_optical_flow_hs(A, B, lam, 6, term_type=ITER)
                check("graph_trim_%d" % i, (u, v), (uo, vo))


def test_plan_query_agrees_with_what_a_solve_reports(hs, gpu_ok):
    """hsflow_plan_query (device-free) and the plan a real solve reports are the same code path."""
    fields = ("kernel", "fuse_steps", "tile_w", "tile_h", "threads", "groups_per_thread", "tiles", "lds_bytes", "jacobi_launches")
    for (W, H, N, it, kw) in ((300, 200, 1, 37, {}), (1920, 1080, 1, 100, {}), (257, 129, 2, 9, dict(kernel=hs.KERNEL_FUSED)),
                              (640, 480, 1, 25, dict(kernel=hs.KERNEL_STRIP, fuse_steps=7)), (500, 400, 1, 12, dict(kernel=hs.KERNEL_SIMPLE)),
                              (320, 240, 1, 20, dict(mode=hs.MODE_CLASSIC, alpha=3.0))):
        want = hs.plan_query(W, H, N, lam=1.0, max_iter=it, term_type=ITER, **kw)
        with hs.HSFlow(W, H, N, own_stream=True) as ctx:
            for i in range(N):
                ctx.set_frames(*synth.random_pair(W, H, seed=i), pair=i)
            got = ctx.solve(lam=1.0, max_iter=it, term_type=ITER, **kw)
        assert {f: got[f] for f in fields} == {f: want[f] for f in fields}, (W, H, N, it, kw)
