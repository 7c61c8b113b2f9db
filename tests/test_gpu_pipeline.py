"""Pair pipeline (hsflow_pipeline_*): host buffers in/out with overlapped copies must give exactly
what the resident-context path gives, pair by pair, whatever the depth or the wait order."""
import numpy as np
import pytest

from opticalflowhs_amd import synth

pytestmark = pytest.mark.gpu

ITER = 1
RMS_TOL = 1e-4  # north_star tolerance on u and v against the oracle


def _pairs(n, W, H):
    out = []
    for i in range(n):
        if i % 2:
            out.append(synth.random_pair(W, H, seed=300 + i))
        else:
            out.append(synth.translating_pair(W, H, seed=300 + i, dx=0.5 + 0.1 * i, dy=-0.25))
    return out


@pytest.mark.parametrize("depth", [1, 2, 3, 5])
def test_pipeline_matches_resident_context_and_oracle(hs, oracle, gpu_ok, depth):
    W, H, it, n = 200, 120, 30, 7
    pairs = _pairs(n, W, H)
    ref = []
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        for A, B in pairs:
            ctx.set_frames(A, B)
            ctx.solve(lam=0.5, max_iter=it, term_type=ITER)
            ref.append(ctx.flow())
    with hs.PairPipeline(W, H, depth=depth) as pl:
        assert pl.depth == depth
        bufs, tickets = [], []
        for A, B in pairs:
            a, b = hs.pinned_empty((H, W), np.uint8), hs.pinned_empty((H, W), np.uint8)
            a[...] = A
            b[...] = B
            u, v = hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)
            u.fill(np.nan)
            v.fill(np.nan)
            tickets.append(pl.submit(a, b, u, v, lam=0.5, max_iter=it))
            bufs.append((u, v))
        assert tickets == list(range(n))
        for t in [n - 1, 0] + list(range(1, n - 1)):  # out of order, some already finished inside submit
            pl.wait(t)
            u, v = bufs[t]
            assert np.array_equal(u, ref[t][0]) and np.array_equal(v, ref[t][1]), t
        pl.drain()
    for (A, B), (u, v) in zip(pairs[:3], bufs[:3]):
        uo, vo = oracle.calc_optical_flow_hs(A, B, 0.5, it, term_type=ITER)
        assert np.sqrt(np.mean((u.astype(np.float64) - uo) ** 2)) <= RMS_TOL
        assert np.sqrt(np.mean((v.astype(np.float64) - vo) ** 2)) <= RMS_TOL


def test_pipeline_pageable_strided_buffers_and_reuse(hs, gpu_ok):
    W, H, it = 130, 70, 12
    pairs = _pairs(4, W, H)
    with hs.HSFlow(W, H, own_stream=True) as ctx, hs.PairPipeline(W, H, depth=2) as pl:
        frames = np.zeros((2, H, W + 9), np.uint8)       # row stride > width
        flow = np.full((2, H, W + 5), np.nan, np.float32)
        for rnd in range(2):                              # the same buffers go round twice
            for A, B in pairs[2 * rnd:2 * rnd + 2]:
                frames[0, :, :W], frames[1, :, :W] = A, B
                t = pl.submit(frames[0, :, :W], frames[1, :, :W], flow[0, :, :W], flow[1, :, :W], lam=2.0, max_iter=it,
                              kernel=hs.KERNEL_FUSED)
                pl.wait(t)
                ctx.set_frames(A, B)
                ctx.solve(lam=2.0, max_iter=it, term_type=ITER)
                u, v = ctx.flow()
                assert np.array_equal(flow[0, :, :W], u) and np.array_equal(flow[1, :, :W], v)
                assert np.isnan(flow[:, :, W:]).all()      # padding columns untouched


def test_pipeline_errors(hs, gpu_ok):
    W, H = 64, 32
    with pytest.raises(hs.HsflowError) as e:
        hs.PairPipeline(W, H, depth=0)
    assert e.value.status == hs._lib.E_ARG
    with pytest.raises(hs.HsflowError) as e:
        hs.PairPipeline(0, H, depth=2)
    assert e.value.status == hs._lib.E_SIZE
    A, B = synth.random_pair(W, H, seed=5)
    u, v = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
    with hs.PairPipeline(W, H, depth=2) as pl:
        with pytest.raises(hs.HsflowError) as e:      # EPS alone needs the host between chunks
            pl.submit(A, B, u, v, lam=1.0, max_iter=5, term_type=hs.TERM_EPS)
        assert e.value.status == hs._lib.E_ARG and "ITER|EPS" in str(e.value)
        with pytest.raises(hs.HsflowError) as e:      # ... and ITER|EPS is asynchronous on the strip kernel only
            pl.submit(A, B, u, v, lam=1.0, max_iter=5, term_type=hs.TERM_ITER | hs.TERM_EPS, kernel=hs.KERNEL_FUSED)
        assert e.value.status == hs._lib.E_ARG
        with pytest.raises(hs.HsflowError) as e:
            pl.wait(0)                                # nothing was issued
        assert e.value.status == hs._lib.E_ARG
        with pytest.raises(ValueError):
            pl.submit(A[:, :-1], B, u, v, max_iter=5)
        with pytest.raises(ValueError):
            pl.submit(A, B, u.astype(np.float64), v, max_iter=5)
        t = pl.submit(A, B, u, v, lam=1.0, max_iter=5)  # still usable after the failures
        assert t == 0
        pl.wait(t)
        pl.wait(t)                                    # idempotent
        assert np.isfinite(u).all() and np.abs(u).max() > 0


def test_pipeline_iter_eps_like_the_reference_call(hs, oracle, gpu_ok):
    """cvCalcOpticalFlowHS(.., ITER|EPS, it, 1e-6) per pair through the pipeline: pairs whose early
    stop never fires stay on the fast path, a pair that converges at once is re-solved inside wait();
    both must equal the synchronous solve and report its sweep count."""
    W, H, it = 260, 140, 45
    eps6 = float(np.float32(1e-6))
    pairs = _pairs(5, W, H)
    flat = np.full((H, W), 77, np.uint8)
    pairs.insert(2, (flat, flat.copy()))               # identical frames: Eps = 0 after the first sweep
    ref = []
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        for A, B in pairs:
            ctx.set_frames(A, B)
            info = ctx.solve(lam=0.3, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
            ref.append((ctx.flow(), info["iterations_done"], info["last_eps"]))
        # the asynchronous form on a plain context: results are final after synchronize()
        ctx.set_frames(*pairs[0])
        ctx.solve_async(lam=0.3, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
        ctx.synchronize()
        i0 = ctx.info()
        assert i0["iterations_done"] == ref[0][1] and i0["eps_rerun"] == 0 and i0["last_eps"] == ref[0][2]
        assert np.array_equal(ctx.flow()[0], ref[0][0][0])
        ctx.set_frames(*pairs[2])
        ctx.solve_async(lam=0.3, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
        u, v = ctx.flow()                               # settles the owed check, then copies
        i2 = ctx.info()
        assert i2["iterations_done"] == 1 and i2["eps_rerun"] == 1 and not u.any() and not v.any()
        # new frames before the owed check was settled: the pending solve is settled first, on ITS frames
        ctx.set_frames(*pairs[2])
        ctx.solve_async(lam=0.3, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
        ctx.set_frames(*pairs[1])
        i3 = ctx.info()
        u, v = ctx.flow()
        assert i3["iterations_done"] == 1 and i3["eps_rerun"] == 1 and not u.any() and not v.any()
        ctx.solve(lam=0.3, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
        assert np.array_equal(ctx.flow()[0], ref[1][0][0])
    assert ref[2][1] == 1 and all(r[1] == it for k, r in enumerate(ref) if k != 2)
    with hs.PairPipeline(W, H, depth=3) as pl:
        outs = []
        for A, B in pairs:
            a, b = hs.pinned_empty((H, W), np.uint8), hs.pinned_empty((H, W), np.uint8)
            a[...], b[...] = A, B
            u, v = hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)
            u.fill(np.nan)
            v.fill(np.nan)
            t = pl.submit(a, b, u, v, lam=0.3, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
            outs.append((t, u, v))
            if t >= 2:                                  # ask while the slot still remembers that pair
                k = t - 2
                info = pl.info(k)
                assert info["iterations_done"] == ref[k][1] and info["eps_rerun"] == (1 if k == 2 else 0), (k, info)
                assert np.array_equal(outs[k][1], ref[k][0][0]) and np.array_equal(outs[k][2], ref[k][0][1]), k
        pl.drain()
        for k in (len(pairs) - 2, len(pairs) - 1):
            assert np.array_equal(outs[k][1], ref[k][0][0]) and np.array_equal(outs[k][2], ref[k][0][1]), k
        with pytest.raises(hs.HsflowError) as e:        # slot long since reused
            pl.info(0)
        assert e.value.status == hs._lib.E_STATE
    uo, vo, n_o, _ = oracle.calc_optical_flow_hs(pairs[1][0], pairs[1][1], 0.3, it, eps6, 3, return_info=True)
    assert n_o == ref[1][1]
    assert np.sqrt(np.mean((outs[1][1].astype(np.float64) - uo) ** 2)) <= RMS_TOL


def test_pipeline_runs_the_cpu_routes_preprocessing_on_the_device(hs, oracle, gpu_ok):
    """Colour frames in, flow out: BGR->gray and the 3x3 blur of OpticalFlowOpenCV::runFromImg ride along in
    each pair's queue.  Must equal the synchronous entry points and the oracle chain, for every format."""
    W, H, it = 150, 90, 14
    eps6 = float(np.float32(1e-6))
    rng = np.random.default_rng(4)
    pairs = [(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8), rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)) for _ in range(5)]
    crit = dict(lam=0.2, max_iter=it, epsilon=eps6, term_type=hs.TERM_ITER | hs.TERM_EPS)
    for frames in ("bgr_blur", "bgr", "gray_blur", "gray"):
        colour = frames.startswith("bgr")
        ref = []
        with hs.HSFlow(W, H, own_stream=True) as ctx:
            for A, B in pairs:
                a, b = (A, B) if colour else (oracle.bgr2gray(A), oracle.bgr2gray(B))
                if frames == "bgr_blur":
                    ctx.set_frames_bgr(a, b, blur=True)
                elif frames == "bgr":
                    ctx.set_frames_bgr(a, b, blur=False)
                elif frames == "gray_blur":
                    ctx.set_frames_gray_blur(a, b)
                else:
                    ctx.set_frames(a, b)
                ctx.solve(**crit)
                ref.append(ctx.flow())
        with hs.PairPipeline(W, H, depth=3) as pl:
            outs = []
            for A, B in pairs:
                src = (A, B) if colour else (oracle.bgr2gray(A), oracle.bgr2gray(B))
                shape = (H, W, 3) if colour else (H, W)
                a, b = hs.pinned_empty(shape, np.uint8), hs.pinned_empty(shape, np.uint8)
                a[...], b[...] = src
                u, v = hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)
                pl.submit(a, b, u, v, frames=frames, **crit)
                outs.append((u, v))
            pl.drain()
        for k, ((u, v), (ur, vr)) in enumerate(zip(outs, ref)):
            assert np.array_equal(u, ur) and np.array_equal(v, vr), (frames, k)
    # the oracle chain for the full route
    A, B = pairs[0]
    ga, gb = oracle.box_blur3(oracle.bgr2gray(A)), oracle.box_blur3(oracle.bgr2gray(B))
    uo, vo = oracle.calc_optical_flow_hs(ga, gb, 0.2, it, eps6, 3)
    with hs.PairPipeline(W, H, depth=2) as pl:
        u, v = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        pl.wait(pl.submit(A, B, u, v, frames="bgr_blur", **crit))
        assert np.sqrt(np.mean((u.astype(np.float64) - uo) ** 2)) <= RMS_TOL and np.sqrt(np.mean((v.astype(np.float64) - vo) ** 2)) <= RMS_TOL
        with pytest.raises(ValueError):
            pl.submit(A[:, :, 0], B, u, v, frames="bgr", **crit)


def test_repeated_async_iter_eps_solves_pass_the_owed_check_on(hs, oracle, gpu_ok):
    """hsflow_solve_async with the reference's ITER|EPS criteria owes an early-stop check.  A following
    asynchronous solve with bit-identical parameters (zero start) takes that check over instead of waiting for
    the stream; whatever settles it in the end must leave exactly what a synchronous solve leaves -- also when
    the witness fails and the exact pass has to run, and when the parameters change in between."""
    import time
    W, H, it = 700, 300, 57
    A, B = synth.translating_pair(W, H, seed=21, dx=1.25, dy=-0.75)
    eps6 = float(np.float32(1e-6))
    EPS = 2
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        want = ctx.solve(lam=0.7, max_iter=it, epsilon=eps6, term_type=ITER | EPS, use_graph=True)
        u0, v0 = ctx.flow()
        p = ctx.make_params(lam=0.7, max_iter=it, epsilon=eps6, term_type=ITER | EPS, use_graph=True)
        for _ in range(3):
            ctx.solve_async(p)
        ctx.synchronize()                       # warm
        t0 = time.perf_counter()
        for _ in range(40):
            ctx.solve_async(p)
        t_enqueue = time.perf_counter() - t0
        ctx.synchronize()
        t_all = time.perf_counter() - t0
        got = ctx.info()
        u, v = ctx.flow()
        # (last_eps: the asynchronous solves measured nothing; info() ran the last launch again to get it)
        assert got["iterations_done"] == it and got["eps_rerun"] == 0 and got["last_eps"] == want["last_eps"]
        assert np.array_equal(u, u0) and np.array_equal(v, v0)
        ctx.solve_async(p)
        assert ctx.info()["last_eps"] == want["last_eps"]
        u, v = ctx.flow()                       # the measuring launch rewrote the same flow
        assert np.array_equal(u, u0) and np.array_equal(v, v0)
        assert t_enqueue < 0.8 * t_all or t_all < 2e-3, (t_enqueue, t_all)   # the host ran ahead of the device
        # witness fails (epsilon above the final Eps): the repeats still end in the exact result
        over = float(want["last_eps"]) * 3.0
        ref = ctx.solve(lam=0.7, max_iter=it, epsilon=over, term_type=ITER | EPS, use_graph=True)
        ur, vr = ctx.flow()
        assert ref["eps_rerun"] == 1 and ref["iterations_done"] < it
        q = ctx.make_params(lam=0.7, max_iter=it, epsilon=over, term_type=ITER | EPS, use_graph=True)
        for _ in range(5):
            ctx.solve_async(q)
        got = ctx.info()                        # settles: exact pass
        u, v = ctx.flow()
        assert got["eps_rerun"] == 1 and got["iterations_done"] == ref["iterations_done"]
        assert np.array_equal(u, ur) and np.array_equal(v, vr)
        # a change of parameters in between settles the old solve first; each result is its own
        ctx.solve_async(p)
        r = ctx.make_params(lam=0.3, max_iter=it, epsilon=eps6, term_type=ITER | EPS, use_graph=True)
        ctx.solve_async(r)
        ctx.solve_async(r)
        u3, v3 = ctx.flow()
        i3 = ctx.info()
        uo, vo = oracle.calc_optical_flow_hs(A, B, 0.3, it, eps6, ITER | EPS)
        assert i3["iterations_done"] == it
        assert np.sqrt(np.mean((u3.astype(np.float64) - uo) ** 2)) <= RMS_TOL and np.sqrt(np.mean((v3.astype(np.float64) - vo) ** 2)) <= RMS_TOL
        # warm starts are never carried over (the state moves on): two use_previous solves = 2 x the sweeps
        ctx.solve(lam=0.3, max_iter=10, term_type=ITER)
        w = ctx.make_params(lam=0.3, max_iter=10, epsilon=eps6, term_type=ITER | EPS, use_previous=True)
        ctx.solve_async(w)
        ctx.solve_async(w)
        u4, v4 = ctx.flow()
        ctx.solve(lam=0.3, max_iter=30, term_type=ITER)
        u5, v5 = ctx.flow()
        assert np.array_equal(u4, u5) and np.array_equal(v4, v5)


@pytest.mark.parametrize("depth", [2, 3])
def test_device_resident_stream_early_stop_while_the_next_pair_is_in_flight(hs, oracle, gpu_ok, depth):
    """The reference's camera loop (/root/reference OpticalFlowOpenCV.cpp:91-95: a fresh pair per step, ITER|EPS) through
    hsflow_pipeline_submit_device: the pairs already lie in device memory, pair k+1 is submitted while pair k's
    early-stop check is still owed.  Pair kind 0 converges inside the budget (its check fails, the pair is re-solved
    from its slot's untouched frames), kind 1 does not.  Every pair's flow and stopping sweep must equal the
    synchronous solve of a plain context, bit for bit, whichever order they are asked for in."""
    import torch
    W, H, budget, lam, eps = 512, 160, 400, 1e-3, 1e-4
    flat_a = np.full((H, W), 90, np.uint8)
    flat_b = flat_a.copy()
    flat_a[40:120, 100:400] = 120
    flat_b[40:120, 100:400] = 121
    kinds = [(flat_a, flat_b), synth.translating_pair(W, H, seed=11), synth.random_pair(W, H, seed=12)]
    ref = []
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        for A, B in kinds:
            ctx.set_frames(A, B)
            i = ctx.solve(lam=lam, max_iter=budget, term_type=3, epsilon=eps)
            ref.append((ctx.flow(), i["iterations_done"]))
    assert 1 < ref[0][1] < budget and ref[1][1] == budget, [r[1] for r in ref]   # kind 0 stops early, kind 1 runs the budget out
    uo, vo, k0, _ = oracle.calc_optical_flow_hs(flat_a, flat_b, lam, budget, epsilon=eps, return_info=True)
    assert abs(k0 - ref[0][1]) <= 1
    dev = [(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()) for A, B in kinds]
    torch.cuda.synchronize()
    order = [0, 1, 0, 2, 1, 0, 0, 2]
    p = hs.make_params(lam=lam, max_iter=budget, term_type=3, epsilon=eps, use_graph=True)
    with hs.PairPipeline(W, H, depth=depth) as pl:
        tickets = [pl.submit_device(dev[k][0], dev[k][1], params=p) if j < depth else None for j, k in enumerate(order)]
        for j, k in enumerate(order):
            # pair j is asked for while pairs j+1 .. j+depth-1 are in flight
            u, v = pl.flow_device(tickets[j])
            info = pl.info(tickets[j])
            assert info["iterations_done"] == ref[k][1] and info["eps_rerun"] == (1 if ref[k][1] < budget else 0), (j, k, info)
            # (bit for bit; below 1e-30 -- reached only on the flat synthetic pair -- the strip kernels' scaled state keeps bits that
            # depend on where the launch boundaries fall, and at depth >= 3 the pipeline picks its own launch shape: DESIGN.md 4.1)
            for got, want in ((u.cpu().numpy(), ref[k][0][0]), (v.cpu().numpy(), ref[k][0][1])):
                assert np.all((got == want) | ((np.abs(got) < 1e-30) & (np.abs(want) < 1e-30))), (j, k)
            if j + depth < len(order):
                kk = order[j + depth]
                tickets[j + depth] = pl.submit_device(dev[kk][0], dev[kk][1], params=p)
        pl.drain()
