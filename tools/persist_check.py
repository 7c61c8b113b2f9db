#!/usr/bin/env python3
"""The persistent launch (HSFLOW_KERNEL_PERSIST) against the launch-per-fuse_steps strip kernel: same bits, and what
a solve costs either way.   usage: tools/persist_check.py [--width W --height H --iters N --fuse-steps T --reps K]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--fuse-steps", type=int, nargs="*", default=[0])
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
W, H, it = args.width, args.height, args.iters
A, B = synth.translating_pair(W, H, seed=args.seed)
eps6 = float(np.float32(1e-6))


def timed(ctx, p, reps):
    for _ in range(30):
        ctx.solve_async(p)
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.solve_async(p)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


with hs.HSFlow(W, H, 1, device=0, own_stream=True) as ctx:
    ctx.set_frames(A, B)
    for T in args.fuse_steps:
        for tt, name in ((hs.TERM_ITER, "ITER"), (hs.TERM_ITER | hs.TERM_EPS, "ITER|EPS")):
            ref = ctx.make_params(lam=1.0, max_iter=it, term_type=tt, epsilon=eps6, kernel=hs.KERNEL_STRIP, fuse_steps=T, use_graph=True)
            per = ctx.make_params(lam=1.0, max_iter=it, term_type=tt, epsilon=eps6, kernel=hs.KERNEL_PERSIST, fuse_steps=T, use_graph=True)
            ctx.solve_async(ref)
            ctx.synchronize()
            i0 = ctx.info()
            u0, v0 = ctx.flow()
            try:
                ctx.solve_async(per)
                ctx.synchronize()
            except hs.HsflowError as e:
                print("T=%d %s: persistent launch refused: %s" % (T, name, e))
                continue
            i1 = ctx.info()
            u1, v1 = ctx.flow()
            same = bool(np.array_equal(u0, u1) and np.array_equal(v0, v1))
            nbad = int(np.count_nonzero(u0 != u1) + np.count_nonzero(v0 != v1))
            t_ref = timed(ctx, ref, args.reps)
            t_per = timed(ctx, per, args.reps)
            i2 = ctx.info()
            print("%dx%d/%d %-8s T=%2d (plan T=%d R=%d thr=%d tiles=%d): strip %.4f ms (%d launches), persistent %.4f ms (%d phases)  %s  "
                  "bit-identical: %s%s  iterations_done %d/%d eps_rerun %d/%d last_eps %.3g/%.3g"
                  % (W, H, it, name, T, i1["fuse_steps"], i1["groups_per_thread"], i1["threads"], i1["tiles"], t_ref, i0["jacobi_launches"], t_per,
                     i1["persistent"], "%+.1f %%" % ((t_per / t_ref - 1) * 100), same, "" if same else " (%d values differ)" % nbad,
                     i0["iterations_done"], i2["iterations_done"], i0["eps_rerun"], i2["eps_rerun"], i0["last_eps"], i2["last_eps"]), flush=True)

# HSFLOW_DEBUG_STAMPS=<file>: one synchronous persistent solve writes its per-workgroup phase totals there ("P" lines)
stamps = os.environ.get("HSFLOW_DEBUG_STAMPS")
if stamps:
    with hs.HSFlow(W, H, 1, device=0, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for T in args.fuse_steps:
            open(stamps, "w").close()
            try:
                ctx.solve(lam=1.0, max_iter=it, term_type=hs.TERM_ITER, kernel=hs.KERNEL_PERSIST, fuse_steps=T)
                i = ctx.solve(lam=1.0, max_iter=it, term_type=hs.TERM_ITER, kernel=hs.KERNEL_PERSIST, fuse_steps=T)
            except hs.HsflowError as e:
                print("stamps: refused:", e)
                continue
            rows = [list(map(int, l.split()[1:])) for l in open(stamps) if l.startswith("P ")]
            rows = np.array(rows[-i["tiles"]:], dtype=np.float64)
            tot = [list(map(int, l.split())) for l in open(stamps) if l and l[0].isdigit()]
            tot = np.array(tot[-i["tiles"]:], dtype=np.float64)
            ghz = float(np.median(tot[:, 4] / (tot[:, 5] * 10.0))) if len(tot) else 2.1  # shader cycles per ns (the real-time clock ticks every 10 ns)
            nb = max(1, i["persistent"] - 1)
            med = lambda c: float(np.median(rows[:, c]))
            for name, col in (("sweeps", 1), ("publish", 2), ("wait", 3), ("reload", 4), ("pre-loop", 6), ("final", 7)):
                x = rows[:, col] / ghz / 1e3
                print("   %-8s total us over the phases: mean %.2f  min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f"
                      % (name, x.mean(), x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max()))
            x = tot[:, 4] / ghz / 1e3
            print("   kernel   us per workgroup: mean %.2f min %.2f max %.2f; first phase load %.2f; sum of the means %.2f"
                  % (x.mean(), x.min(), x.max(), (tot[:, 1] / ghz / 1e3).mean(), (tot[:, 1] / ghz / 1e3).mean() + sum((rows[:, c] / ghz / 1e3).mean() for c in (1, 2, 3, 4))))
            print("stamps T=%d phases=%d clock %.2f GHz: load0 %.2f us, sweeps %.2f us (%.0f cycles per sweep), per boundary: publish %.2f us, wait %.2f us (max %.2f), reload %.2f us; total %.2f us"
                  % (i["fuse_steps"], i["persistent"], ghz, np.median(tot[:, 1]) / ghz / 1e3, med(1) / ghz / 1e3, med(1) / it,
                     med(2) / nb / ghz / 1e3, med(3) / nb / ghz / 1e3, float(np.max(rows[:, 3])) / nb / ghz / 1e3, med(4) / nb / ghz / 1e3,
                     np.median(tot[:, 4]) / ghz / 1e3))
