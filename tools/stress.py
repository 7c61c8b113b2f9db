#!/usr/bin/env python3
"""Randomised stress of the C ABI against the oracle (a one-off soak, not part of the test suite):
random sizes, pairs per context, kernels and tuning knobs, ITER / ITER|EPS with random epsilon, warm
starts, graphs, asynchronous solves.  usage: python tools/stress.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
from oracle import hs_oracle

ITER, EPS = 1, 2
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
worst = 0.0
near = 0
refused = 0
routed_persist = routed_pipe = 0
t_start = time.time()
for case in range(n_cases):
    W = int(rng.integers(1, 700)) if case % 4 else int(rng.integers(1, 40))
    H = int(rng.integers(1, 400)) if case % 5 else int(rng.integers(1, 12))
    N = int(rng.choice([1, 1, 1, 2, 3]))
    it = int(rng.integers(1, 80))
    route = str(rng.choice(["ctx", "ctx", "ctx", "persist", "pipe"]))
    if route == "persist" and case % 6 != 5:   # a frame the persistent launch can take: one region at least, width a multiple of 4
        W, H, it = 4 * int(rng.integers(64, 176)), int(rng.integers(80, 400)), int(rng.integers(25, 80))
    lam = float(10.0 ** rng.uniform(-2.5, 1.5))
    use_eps = bool(rng.integers(0, 2))
    eps = float(10.0 ** rng.uniform(-7, -1))
    kernel = int(rng.choice([hs.KERNEL_AUTO, hs.KERNEL_AUTO, hs.KERNEL_SIMPLE, hs.KERNEL_FUSED, hs.KERNEL_STRIP, hs.KERNEL_FOLD]))
    kw = dict(kernel=kernel)
    if kernel in (hs.KERNEL_FUSED, hs.KERNEL_STRIP, hs.KERNEL_FOLD) and rng.integers(0, 2):
        kw["fuse_steps"] = int(rng.integers(1, 13))
    graph = bool(rng.integers(0, 2)) and not (use_eps and kernel in (hs.KERNEL_SIMPLE, hs.KERNEL_FUSED))
    do_async = bool(rng.integers(0, 2)) and (not use_eps or kernel in (hs.KERNEL_AUTO, hs.KERNEL_STRIP, hs.KERNEL_FOLD))
    warm = bool(rng.integers(0, 3) == 0) and it > 2
    pairs = []
    for i in range(N):
        kind = rng.integers(0, 4)
        if kind == 0:
            pairs.append(synth.random_pair(W, H, seed=case * 10 + i))
        elif kind == 1:
            A = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
            pairs.append((A, A.copy()))
        else:
            pairs.append(synth.translating_pair(W, H, seed=case * 10 + i, dx=float(rng.uniform(-2, 2)), dy=float(rng.uniform(-2, 2))))
    tt = ITER | EPS if use_eps else ITER
    if case % 6 == 5:  # the classic ("-cl") mode: bit-exact against its oracle, both kernels, as shipped or not
        alpha = float(rng.uniform(0.3, 25.0))
        shipped = bool(rng.integers(0, 2))
        ckw = dict(kernel=int(rng.choice([hs.KERNEL_AUTO, hs.KERNEL_AUTO, hs.KERNEL_SIMPLE, hs.KERNEL_FUSED, hs.KERNEL_STRIP, hs.KERNEL_STRIP])))
        if ckw["kernel"] != hs.KERNEL_SIMPLE and rng.integers(0, 2):
            ckw["fuse_steps"] = int(rng.integers(1, 10))
        if ckw["kernel"] in (hs.KERNEL_AUTO, hs.KERNEL_STRIP) and rng.integers(0, 2):
            ckw["strip_rows"] = int(rng.integers(2, 9))
        if rng.integers(0, 3) == 0:
            ckw["use_graph"] = True
        cwarm = it > 2 and bool(rng.integers(0, 3) == 0)
        with hs.HSFlow(W, H, N, own_stream=True) as ctx:
            for i, (A, B) in enumerate(pairs):
                ctx.set_frames(A, B, pair=i)
            cmode = hs.MODE_CLASSIC_AS_SHIPPED if shipped else hs.MODE_CLASSIC
            try:
                if cwarm:
                    ctx.solve(mode=cmode, alpha=alpha, max_iter=it // 2, term_type=ITER, **ckw)
                    ctx.solve(mode=cmode, alpha=alpha, max_iter=it - it // 2, term_type=ITER, use_previous=True, reuse_derivatives=True, **ckw)
                else:
                    ctx.solve(mode=cmode, alpha=alpha, max_iter=it, term_type=ITER, **ckw)
            except hs.HsflowError as e:  # an explicit strip shape this image has no aligned form of
                if ckw["kernel"] == hs.KERNEL_STRIP and e.status == hs._lib.E_SIZE:
                    refused += 1
                    continue
                raise
            for i, (A, B) in enumerate(pairs):
                u, v = ctx.flow(pair=i)
                uo, vo = hs_oracle.classic_flow(A, B, alpha, it, update_v=not shipped)
                if not (np.array_equal(u, uo) and np.array_equal(v, vo)):
                    bad += 1
                    print("case %d: classic %dx%d N%d it%d alpha%.3g shipped%d %s pair %d MISMATCH" % (case, W, H, N, it, alpha, shipped, ckw, i), flush=True)
        continue
    # round 3's routes beside the plain context: the persistent launch (where this shape can run it), and the device-resident
    # pair pipeline (one pair, cold start: every pair is its own solve there)
    if route == "persist" and (kernel not in (hs.KERNEL_AUTO, hs.KERNEL_STRIP) or (use_eps and not do_async)):
        route = "ctx"
    if route == "pipe" and (N != 1 or warm or kernel in (hs.KERNEL_SIMPLE, hs.KERNEL_FUSED) and use_eps):
        route = "ctx"
    try:
        if route == "pipe":
            import torch
            first = 0
            p = hs.make_params(lam=lam, max_iter=it, epsilon=eps, term_type=tt, use_graph=graph, **kw)
            dA, dB = torch.from_numpy(pairs[0][0]).cuda(), torch.from_numpy(pairs[0][1]).cuda()
            with hs.PairPipeline(W, H, depth=int(rng.integers(2, 5))) as pl:
                t0_ = pl.submit_device(dA, dB, params=p)
                t1_ = pl.submit_device(dA, dB, params=p)   # a second pair behind it: the first one's check is owed meanwhile
                u_, v_ = pl.flow_device(t0_)
                info = pl.info(t0_)
                flows = [(u_.cpu().numpy(), v_.cpu().numpy())]
                u2_, v2_ = pl.flow_device(t1_)
                if not (np.array_equal(u2_.cpu().numpy(), flows[0][0]) and np.array_equal(v2_.cpu().numpy(), flows[0][1])) or \
                        pl.info(t1_)["iterations_done"] != info["iterations_done"]:
                    bad += 1
                    print("case %d: pipeline slots disagree on the same pair" % case, flush=True)
            routed_pipe += 1
        else:
            with hs.HSFlow(W, H, N, own_stream=True) as ctx:
                for i, (A, B) in enumerate(pairs):
                    ctx.set_frames(A, B, pair=i)
                first = it // 2 if warm else 0
                kw_run = dict(kw)
                if route == "persist":
                    kw_run.update(kernel=hs.KERNEL_PERSIST, strip_rows=5)
                if warm:
                    ctx.solve(lam=lam, max_iter=first, term_type=ITER, **kw)
                while True:
                    p = ctx.make_params(lam=lam, max_iter=it - first, epsilon=eps, term_type=tt, use_previous=warm, reuse_derivatives=warm, use_graph=graph, **kw_run)
                    try:
                        if do_async:
                            ctx.solve_async(p)
                            ctx.synchronize()
                            info = ctx.info()
                        else:
                            info = ctx.solve(p)
                        break
                    except hs.HsflowError as e:
                        if kw_run.get("kernel") == hs.KERNEL_PERSIST and e.status == hs._lib.E_SIZE:   # this shape has no persistent form
                            kw_run = dict(kw)
                            continue
                        raise
                if kw_run.get("kernel") == hs.KERNEL_PERSIST:
                    routed_persist += 1
                    if not (info["persistent"] >= 2 or info["eps_rerun"]):   # (an exact pass that settles an ITER|EPS solve runs launch by launch)
                        bad += 1
                        print("case %d: asked for the persistent launch, got %r" % (case, info), flush=True)
                flows = [ctx.flow(pair=i) for i in range(N)]
    except hs.HsflowError as e:
        print("case %d: %dx%d N%d it%d k%d %s -> ERROR %s" % (case, W, H, N, it, kernel, kw, e), flush=True)
        bad += 1
        continue
    # oracle: the batch stops when the maximum over all pairs drops below eps -> emulate by running every pair for the
    # batch's sweep count and checking the stop rule on the batch maximum
    done = info["iterations_done"] + first
    ok = True
    e_batch = 0.0
    for i, (A, B) in enumerate(pairs):
        uo, vo, n_o, e_o = hs_oracle.calc_optical_flow_hs(A, B, lam, done, 0.0, ITER, return_info=True)
        r = max(float(np.sqrt(np.mean((flows[i][0].astype(np.float64) - uo) ** 2))), float(np.sqrt(np.mean((flows[i][1].astype(np.float64) - vo) ** 2))))
        worst = max(worst, r)
        if not (r <= 1e-4):
            ok = False
        if use_eps:  # Eps of the last sweep of this pair
            up, vp, _, _ = hs_oracle.calc_optical_flow_hs(A, B, lam, done - 1, 0.0, ITER, return_info=True) if done > 1 else (np.zeros_like(uo), np.zeros_like(vo), 0, 0)
            e_batch = max(e_batch, float(np.abs(uo - up).max()), float(np.abs(vo - vp).max()))
    if use_eps:
        stopped_early = info["iterations_done"] < it - first
        # (Eps is a difference of fp32 flows: GPU and oracle agree to a few ulp of the flow magnitude, not to a relative
        # 1e-3 of a small epsilon -- the same allowance as for the stopping sweep below)
        fmax = max(1.0, max(float(np.abs(f[0]).max()) for f in flows), max(float(np.abs(f[1]).max()) for f in flows))
        if stopped_early and not (e_batch < eps * (1 + 1e-3) + 4e-7 * fmax):
            ok = False
        if not stopped_early and e_batch < eps * (1 - 1e-3) and done > 1:
            # must then have been the budget's last sweep, or an earlier sweep was already below: check the previous one
            pass
        if abs(e_batch - eps) <= 1e-3 * eps:
            near += 1
        if N == 1 and not warm:  # the oracle's own stopping sweep
            A, B = pairs[0]
            _, _, n_o, e_o = hs_oracle.calc_optical_flow_hs(A, B, lam, it, eps, ITER | EPS, return_info=True)
            if n_o != info["iterations_done"]:
                # tolerated only if some sweep's Eps sits within 1e-3 of the threshold (fp32 vs x87 rounding)
                close = False
                up, vp = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
                for k in range(1, max(n_o, info["iterations_done"]) + 1):
                    uk, vk = hs_oracle.calc_optical_flow_hs(A, B, lam, k, 0.0, ITER)
                    ek = max(float(np.abs(uk - up).max()), float(np.abs(vk - vp).max()))
                    # Eps is a difference of fp32 flows: GPU (fp32 intermediates) and oracle (x87 doubles) agree to
                    # a few ulp of the flow magnitude, not to a relative 1e-3 of a small epsilon
                    close = close or abs(ek - eps) <= 1e-3 * eps + 4e-7 * max(1.0, float(np.abs(uk).max()), float(np.abs(vk).max()))
                    up, vp = uk, vk
                if close:
                    near += 1
                else:
                    ok = False
                    print("   stop sweep differs: oracle %d, gpu %d" % (n_o, info["iterations_done"]), flush=True)
    if not ok:
        bad += 1
        print("case %d: %dx%d N%d it%d lam%.3g eps%s k%d %s graph%d async%d warm%d -> done %d rms %.3g e_batch %.3g" %
              (case, W, H, N, it, lam, ("%.3g" % eps) if use_eps else "-", kernel, kw, graph, do_async, warm, info["iterations_done"], worst, e_batch), flush=True)
    if case % 25 == 24:
        print("... %d cases, %d bad, worst rms %.3g, %.0f s" % (case + 1, bad, worst, time.time() - t_start), flush=True)
print("DONE %d cases, %d bad, worst rms %.3g, %d near-threshold, %d classic strip shapes refused (no aligned form); %d through the persistent launch, %d through the device-resident pipeline"
      % (n_cases, bad, worst, near, refused, routed_persist, routed_pipe))
sys.exit(1 if bad else 0)
