#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timing of the strip kernel from in-kernel clock stamps
(HSFLOW_DEBUG_STAMPS).  Reports load / sweeps / store cycles per workgroup and the shader clock.
This build path is for diagnosis only; its run time is never quoted as a benchmark."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--kernel", default="strip", choices=["strip", "fold"])
    ap.add_argument("--configs", default="10:4:1024;10:5:768;12:4:1024;5:4:1024;8:8:512", help="T:R:threads;...")
    args = ap.parse_args()
    path = tempfile.mktemp(prefix="hs_stamps_")
    os.environ["HSFLOW_DEBUG_STAMPS"] = path
    import opticalflowhs_amd as hs
    from opticalflowhs_amd import synth
    ctx = hs.HSFlow(args.width, args.height, args.pairs, own_stream=True)
    for i in range(args.pairs):
        A, B = synth.translating_pair(args.width, args.height, seed=1 + i)
        ctx.set_frames(A, B, pair=i)
    KERN = hs.KERNEL_FOLD if args.kernel == "fold" else hs.KERNEL_STRIP
    for cfg in args.configs.split(";"):
        T, R, nt = [int(x) for x in cfg.split(":")]
        for _ in range(3):  # warm
            ctx.solve(lam=1.0, max_iter=4 * T, term_type=hs.TERM_ITER, kernel=KERN, fuse_steps=T, strip_rows=R, threads=nt)
        if os.path.exists(path):
            os.remove(path)
        info = ctx.solve(lam=1.0, max_iter=4 * T, term_type=hs.TERM_ITER, kernel=KERN, fuse_steps=T, strip_rows=R, threads=nt)
        rows = np.array([[int(x) for x in l.split()] for l in open(path) if not l.startswith("#")], dtype=np.float64)
        load, sweeps, store, total, rt, xcc = rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4], rows[:, 5], rows[:, 6]
        ghz = total.sum() / (rt.sum() * 10.0)
        print("T=%d R=%d threads=%d tiles=%d tile=%dx%d: clock %.2f GHz" % (T, R, nt, info["tiles"], info["tile_w"], info["tile_h"], ghz))
        for name, a in (("load", load), ("sweeps", sweeps), ("store", store), ("total", total)):
            print("   %-7s cycles: median %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f   (median %.2f us; per sweep %.0f cyc)"
                  % (name, np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.max(), np.median(a) / ghz / 1e3,
                     np.median(a) / T if name == "sweeps" else 0))
        print("   wall per workgroup (100 MHz ticks): median %.2f us, max %.2f us; xcc ids seen: %s"
              % (np.median(rt) / 100.0, rt.max() / 100.0, sorted(set(int(x) & 0xF for x in xcc))))
    ctx.close()
    if os.path.exists(path):
        os.remove(path)


if __name__ == "__main__":
    main()
