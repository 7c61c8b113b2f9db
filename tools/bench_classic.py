#!/usr/bin/env python3
"""Rate of the classic ("-cl" route, Kernels.cl semantics) mode: one JSON line.
usage: python tools/bench_classic.py [--width 1920 --height 1080 --iters 100 --steps 50]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--alpha", type=float, default=15.0)
    ap.add_argument("--kernel", default="auto", choices=["auto", "simple", "fused", "strip"])
    ap.add_argument("--fuse", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    import opticalflowhs_amd as hs
    from opticalflowhs_amd import synth
    W, H = args.width, args.height
    A, B = synth.translating_pair(W, H, seed=1)
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        kw = dict(kernel={"auto": hs.KERNEL_AUTO, "simple": hs.KERNEL_SIMPLE, "fused": hs.KERNEL_FUSED, "strip": hs.KERNEL_STRIP}[args.kernel],
                  fuse_steps=args.fuse, strip_rows=args.rows, threads=args.threads)
        p = ctx.make_params(mode=hs.MODE_CLASSIC, alpha=args.alpha, max_iter=args.iters, term_type=hs.TERM_ITER, **kw)
        for _ in range(5):
            ctx.solve_async(p)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.solve_async(p)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        pi = ctx.solve(ctx.make_params(mode=hs.MODE_CLASSIC, alpha=args.alpha, max_iter=args.iters, term_type=hs.TERM_ITER, profile=True, **kw))
    px = W * H
    print(json.dumps({"mode": "classic", "kernel": pi["kernel"], "fuse_steps": pi["fuse_steps"], "rows": pi["groups_per_thread"], "threads": pi["threads"],
                      "tile": [pi["tile_w"], pi["tile_h"]], "launches": pi["jacobi_launches"], "width": W, "height": H, "iters": args.iters, "ms_per_solve": round(dt * 1e3, 4),
                      "mpix_iter_per_s": round(px * args.iters / dt / 1e6), "us_per_sweep_kernel": round(pi["jacobi_ms"] * 1e3 / args.iters, 3),
                      "alg_GBps_28B": round(28.0 * px * args.iters / (pi["jacobi_ms"] * 1e-3) / 1e9, 1)}))


if __name__ == "__main__":
    main()
