#!/usr/bin/env python3
"""Classic mode, register-strip kernel: time of a solve over (rows per lane, threads, sweeps per launch).
usage: python tools/sweep_classic_strip.py [--width 1920 --height 1080 --iters 100] [--configs "R:threads:T;..."]"""
import argparse
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--configs", default="")
    ap.add_argument("--graph", action="store_true", help="the solve as one hipGraph")
    args = ap.parse_args()
    import opticalflowhs_amd as hs
    from opticalflowhs_amd import synth
    W, H = args.width, args.height
    A, B = synth.translating_pair(W, H, seed=1)
    if args.configs:
        cfgs = [tuple(int(x) for x in c.split(":")) for c in args.configs.split(";")]
    else:
        cfgs = [(R, nt, T) for R, nts in ((3, (1024,)), (4, (768, 512)), (5, (768, 512)), (6, (512,)), (8, (512, 256)))
                for nt in nts for T in (6, 8, 10, 12, 16, 20)]
    print("rows threads T  tile  tiles launches  ms_per_solve  us_per_sweep")
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for kern, R, nt, T in [(hs.KERNEL_FUSED, 0, 0, 0)] + [(hs.KERNEL_STRIP,) + c for c in cfgs]:
            try:
                p = ctx.make_params(mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=args.iters, term_type=hs.TERM_ITER, kernel=kern,
                                    strip_rows=R, threads=nt, fuse_steps=T, use_graph=args.graph)
                for _ in range(3):
                    ctx.solve_async(p)
                ctx.synchronize()
            except hs.HsflowError as e:
                print(R, nt, T, "refused:", e)
                continue
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx.solve_async(p)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            i = ctx.info()
            print("%d %4d %2d  %dx%d  %d %d  %.4f  %.3f" % (i["groups_per_thread"], i["threads"], i["fuse_steps"], i["tile_w"], i["tile_h"],
                                                          i["tiles"], i["jacobi_launches"], dt * 1e3, dt * 1e6 / args.iters), flush=True)


if __name__ == "__main__":
    main()
