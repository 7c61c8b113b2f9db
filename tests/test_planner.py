"""The launch planners, queried without a device (hsflow_plan_query runs the solver's own planning code):
every plan must respect the limits the kernels were compiled for and cover the frame exactly once."""
import numpy as np
import pytest

LDS_LIMIT = 160 * 1024


def ceil_div(a, b):
    return -(-a // b)


def check_plan(hs, W, H, N, it, info, kw):
    k = info["kernel"]
    T = info["fuse_steps"]
    assert info["width"] == W and info["height"] == H and info["n_pairs"] == N and info["pitch"] % 64 == 0 and info["pitch"] >= W
    if k == hs.KERNEL_SIMPLE:
        assert T == 1 and info["jacobi_launches"] == it
        return
    assert 1 <= T <= 32
    if info["persistent"]:   # the whole budget as one launch in phases of T sweeps (HSFLOW_KERNEL_PERSIST): one tile per CU at most
        assert k == hs.KERNEL_STRIP and info["jacobi_launches"] == 1 and info["persistent"] == ceil_div(it, T) >= 2
        assert info["tiles"] <= 256 and W % 4 == 0 and info["tile_h"] >= T
    else:
        assert info["jacobi_launches"] == ceil_div(it, T)
    assert info["tile_w"] >= 1 and info["tile_h"] >= 1
    assert info["tiles"] == ceil_div(W, info["tile_w"]) * ceil_div(H, info["tile_h"]) * N      # core tiles cover the frame once
    assert info["threads"] % 64 == 0 and 64 <= info["threads"] <= 1024
    assert 0 < info["lds_bytes"] <= LDS_LIMIT
    hx = ceil_div(T, 4) * 4
    if k in (hs.KERNEL_STRIP, hs.KERNEL_FOLD):
        R, NW = info["groups_per_thread"], info["threads"] // 64
        fold = k == hs.KERNEL_FOLD
        assert 1 <= R <= 8
        # the register budget each instantiation was compiled for (launch bounds): waves per workgroup
        reff = R + 1 if fold else R
        assert NW <= (16 if reff <= 5 else 12 if reff <= 6 else 8), (R, NW, fold)
        assert info["tile_w"] == (128 if fold else 256) - 2 * hx                                 # region width = one wavefront
        assert info["tile_h"] == NW * R * (2 if fold else 1) - 2 * T                             # rows held minus the halo
    else:  # LDS tile: region = core + halo must fit the lanes and the LDS
        K, NT = info["groups_per_thread"], info["threads"]
        rw4 = (info["tile_w"] + 2 * hx) // 4
        rh = info["tile_h"] + 2 * T
        assert info["tile_w"] % 4 == 0 and NT in (256, 512, 1024) and 1 <= K <= (3 if NT == 1024 else 4)
        assert rw4 * rh <= NT * K
        assert 2 * (4 * rw4 + 8) * (rh + 2) * 4 == info["lds_bytes"]


def test_plans_respect_kernel_limits_and_cover_the_frame(hs):
    rng = np.random.default_rng(17)
    seen = set()
    for case in range(400):
        W = int(rng.integers(1, 5000)) if case % 3 else int(rng.integers(1, 64))
        H = int(rng.integers(1, 3000)) if case % 4 else int(rng.integers(1, 40))
        N = int(rng.choice([1, 1, 2, 7, 64]))
        it = int(rng.integers(1, 600))
        kernel = int(rng.choice([hs.KERNEL_AUTO, hs.KERNEL_AUTO, hs.KERNEL_SIMPLE, hs.KERNEL_FUSED, hs.KERNEL_STRIP, hs.KERNEL_FOLD]))
        kw = dict(kernel=kernel)
        if rng.integers(0, 3) == 0 and kernel != hs.KERNEL_SIMPLE:
            kw["fuse_steps"] = int(rng.integers(1, 25))
        tt = int(rng.choice([hs.TERM_ITER, hs.TERM_ITER | hs.TERM_EPS]))
        try:
            info = hs.plan_query(W, H, N, lam=1.0, max_iter=it, term_type=tt, **kw)
        except hs.HsflowError as e:
            assert e.status == hs._lib.E_SIZE and "no feasible" in str(e), (W, H, N, it, kw, e)   # e.g. a halo wider than the region
            continue
        check_plan(hs, W, H, N, it, info, kw)
        seen.add(info["kernel"])
        if kernel == hs.KERNEL_AUTO:   # the AUTO rule
            assert info["kernel"] == (hs.KERNEL_FOLD if W * H * N <= 1500000 else hs.KERNEL_STRIP)
    assert seen == {hs.KERNEL_SIMPLE, hs.KERNEL_FUSED, hs.KERNEL_STRIP, hs.KERNEL_FOLD}


def test_planner_known_configs_and_refusals(hs):
    c2 = hs.plan_query(1920, 1080, lam=1.0, max_iter=100, term_type=hs.TERM_ITER)           # BASELINE config C2
    assert c2["kernel"] == hs.KERNEL_STRIP and c2["tiles"] <= 256 and c2["jacobi_launches"] <= 9    # one round of workgroups
    c3 = hs.plan_query(3840, 2160, lam=1.0, max_iter=200, term_type=hs.TERM_ITER)           # C3
    assert c3["kernel"] == hs.KERNEL_STRIP
    cl = hs.plan_query(1920, 1080, mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=100, term_type=hs.TERM_ITER)
    assert cl["kernel"] == hs.KERNEL_STRIP and cl["tiles"] <= 256 and cl["jacobi_launches"] <= 17   # register strip, one round
    cf = hs.plan_query(1920, 1080, mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=100, term_type=hs.TERM_ITER, kernel=hs.KERNEL_FUSED)
    assert cf["kernel"] == hs.KERNEL_FUSED and cf["fuse_steps"] == 6 and cf["jacobi_launches"] == 17
    # no strip shape (more sweeps per launch than a region has halo rows for) or an alpha outside the range of the strip
    # kernel's division: AUTO falls back to the LDS tile, an explicit request is refused
    for kw in (dict(width=200, height=37, alpha=15.0, fuse_steps=30), dict(width=1920, height=1080, alpha=1e-9)):
        w, h = kw.pop("width"), kw.pop("height")
        assert hs.plan_query(w, h, mode=hs.MODE_CLASSIC, max_iter=20, term_type=hs.TERM_ITER, **kw)["kernel"] == hs.KERNEL_FUSED
        with pytest.raises(hs.HsflowError):
            hs.plan_query(w, h, mode=hs.MODE_CLASSIC, max_iter=20, term_type=hs.TERM_ITER, kernel=hs.KERNEL_STRIP, **kw)
    for bad, status in ((dict(width=0, height=5), hs._lib.E_SIZE), (dict(width=5, height=5, n_pairs=0), hs._lib.E_SIZE)):
        with pytest.raises(hs.HsflowError) as e:
            hs.plan_query(bad.get("width"), bad.get("height"), bad.get("n_pairs", 1), max_iter=5)
        assert e.value.status == status
    with pytest.raises(hs.HsflowError) as e:
        hs.plan_query(64, 64, lam=-1.0, max_iter=5)
    assert e.value.status == hs._lib.E_ARG
    with pytest.raises(hs.HsflowError) as e:
        hs.plan_query(64, 64, max_iter=0, term_type=hs.TERM_ITER)
    assert e.value.status == hs._lib.E_NOTERM
    with pytest.raises(hs.HsflowError) as e:
        hs.plan_query(64, 64, max_iter=5, kernel=99)
    assert e.value.status == hs._lib.E_ARG


def test_eps_without_budget_needs_a_usable_epsilon(hs):
    """EPS alone (or ITER|EPS with max_iter <= 0) stops only on Eps < epsilon: epsilon <= 0, NaN or inf would never
    stop (the original spins forever, cv210.dll VA 0x1012f10b-0x1012f14a) -> HSFLOW_E_NOTERM."""
    for tt, it in ((hs.TERM_EPS, 0), (hs.TERM_EPS, 50), (hs.TERM_ITER | hs.TERM_EPS, 0), (hs.TERM_ITER | hs.TERM_EPS, -3)):
        for eps in (0.0, -1e-6, float("nan"), float("inf")):
            with pytest.raises(hs.HsflowError) as e:
                hs.plan_query(64, 64, max_iter=it, term_type=tt, epsilon=eps)
            assert e.value.status == hs._lib.E_NOTERM, (tt, it, eps)
        assert hs.plan_query(64, 64, max_iter=it, term_type=tt, epsilon=1e-6)["kernel"] in (hs.KERNEL_FOLD, hs.KERNEL_STRIP)
    # with a sweep budget any epsilon is fine: ITER ends the solve
    for eps in (0.0, -1.0, float("nan")):
        assert hs.plan_query(64, 64, max_iter=7, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=eps)["jacobi_launches"] >= 1
