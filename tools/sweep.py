#!/usr/bin/env python3
"""Tile / fuse-depth sweep of the fused Jacobi kernel (BASELINE config C3 asks for an LDS
tile-size sweep).  Every variant runs in ONE process, interleaved over several rounds; the table
reports the median and minimum time per solve.  Results go to gpurun_out/sweep_<tag>.csv.

usage: python tools/sweep.py --width 1920 --height 1080 --iters 100 [--pairs 1] [--tag 1080p]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tag", default="sweep")
    ap.add_argument("--fuse", default="4,5,8,10,20")
    ap.add_argument("--rw4", default="8,12,16,20,24,32,40,48,64")
    ap.add_argument("--threads", default="256,512,1024")
    ap.add_argument("--kmax", type=int, default=4)
    ap.add_argument("--extra", default="", help="semicolon list of T:tile_w:tile_h:threads")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--strip", default="", help="strip-kernel sweep: comma list of rows per lane, e.g. 1,2,3,4,5,6,7,8")
    ap.add_argument("--waves", default="4,6,8,10,12,14,16", help="strip-kernel sweep: wavefronts per workgroup")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--fold", default="", help="fold-kernel sweep: comma list of rows per lane")
    args = ap.parse_args()

    import torch
    import opticalflowhs_amd as hs
    from opticalflowhs_amd import synth

    W, H, iters = args.width, args.height, args.iters
    ctx = hs.HSFlow(W, H, args.pairs, own_stream=True)
    for i in range(args.pairs):
        A, B = synth.translating_pair(W, H, seed=1 if args.pairs == 1 else 1000 + i)
        ctx.set_frames(A, B, pair=i)

    variants = [("auto", dict()), ("simple", dict(kernel=hs.KERNEL_SIMPLE)), ("fused_auto", dict(kernel=hs.KERNEL_FUSED)),
                ("strip_auto", dict(kernel=hs.KERNEL_STRIP)), ("fold_auto", dict(kernel=hs.KERNEL_FOLD))]
    for T in [int(x) for x in args.fuse.split(",") if x]:
        for R in [int(x) for x in args.strip.split(",") if x]:
            for NW in [int(x) for x in args.waves.split(",") if x]:
                if NW * R - 2 * T < 1 or NW > (16 if R <= 5 else 12 if R == 6 else 8):
                    continue
                variants.append(("S_T%d_R%d_NW%d" % (T, R, NW),
                                 dict(kernel=hs.KERNEL_STRIP, fuse_steps=T, strip_rows=R, threads=NW * 64)))
    for T in [int(x) for x in args.fuse.split(",") if x]:
        for R in [int(x) for x in args.fold.split(",") if x]:
            for NW in [int(x) for x in args.waves.split(",") if x]:
                if NW * R * 2 - 2 * T < 1 or NW > (16 if R <= 4 else 12 if R == 5 else 8):
                    continue
                variants.append(("F_T%d_R%d_NW%d" % (T, R, NW),
                                 dict(kernel=hs.KERNEL_FOLD, fuse_steps=T, strip_rows=R, threads=NW * 64)))
    for T in [int(x) for x in args.fuse.split(",") if x]:
        if iters % T or args.no_fused:
            continue
        HX = (T + 3) // 4 * 4
        for NT in [int(x) for x in args.threads.split(",")]:
            for K in range(1, args.kmax + 1):
                if NT == 1024 and K > 3:
                    continue
                for RW4 in [int(x) for x in args.rw4.split(",")]:
                    CW = 4 * RW4 - 2 * HX
                    RH = NT * K // RW4
                    CH = RH - 2 * T
                    if CW < 4 or CH < 1:
                        continue
                    if 2 * (4 * RW4 + 8) * (RH + 2) * 4 > 160 * 1024:
                        continue
                    if (RW4 * RH + NT - 1) // NT != K:
                        continue
                    variants.append(("T%d_%dx%d_nt%d_k%d" % (T, CW, CH, NT, K),
                                     dict(kernel=hs.KERNEL_FUSED, fuse_steps=T, tile_w=CW, tile_h=CH, threads=NT)))
    for e in [x for x in args.extra.split(";") if x]:
        T, tw, th, nt = [int(v) for v in e.split(":")]
        variants.append(("X_T%d_%dx%d_nt%d" % (T, tw, th, nt),
                         dict(kernel=hs.KERNEL_FUSED, fuse_steps=T, tile_w=tw, tile_h=th, threads=nt)))

    params, infos, ok = [], [], []
    for name, kw in variants:
        p = ctx.make_params(lam=1.0, max_iter=iters, term_type=hs.TERM_ITER, use_graph=not args.no_graph, **kw)
        try:
            ctx.solve_async(p)
            ctx.synchronize()
            infos.append(ctx.info())
            params.append(p)
            ok.append(name)
        except hs.HsflowError as e:
            print("skip %s: %s" % (name, e))
    times = [[] for _ in ok]
    for r in range(args.rounds):
        for i, p in enumerate(params):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                ctx.solve_async(p)
            ctx.synchronize()
            times[i].append((time.perf_counter() - t0) / args.reps * 1e3)
    px = W * H * args.pairs * iters
    rows = []
    for i, name in enumerate(ok):
        med, mn = float(np.median(times[i])), float(np.min(times[i]))
        inf = infos[i]
        rows.append((med, name, mn, px / med / 1e3, inf["tiles"], inf["lds_bytes"], inf["groups_per_thread"],
                     inf["threads"], inf["tile_w"], inf["tile_h"], inf["fuse_steps"]))
    rows.sort()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "sweep_%s.csv" % args.tag)
    with open(path, "w") as f:
        f.write("name,median_ms,min_ms,mpix_iter_per_s,tiles,lds_bytes,k,threads,tile_w,tile_h,T\n")
        for med, name, mn, rate, tiles, lds, k, nt, tw, th, T in rows:
            f.write("%s,%.4f,%.4f,%.0f,%d,%d,%d,%d,%d,%d,%d\n" % (name, med, mn, rate, tiles, lds, k, nt, tw, th, T))
    print("%d variants, %dx%d x%d pairs, %d iters; best 15:" % (len(rows), W, H, args.pairs, iters))
    for med, name, mn, rate, tiles, lds, k, nt, tw, th, T in rows[:15]:
        print("%-28s med %.4f ms  min %.4f ms  %9.0f Mpix*it/s  tiles %4d lds %6d" % (name, med, mn, rate, tiles, lds))
    for med, name, mn, rate, tiles, lds, k, nt, tw, th, T in rows:
        if name in ("auto", "simple", "fused_auto", "strip_auto", "fold_auto"):
            print("%-28s med %.4f ms  min %.4f ms  %9.0f Mpix*it/s  tiles %4d (T %d tile %dx%d nt %d)" % (name, med, mn, rate, tiles, T, tw, th, nt))
    ctx.close()


if __name__ == "__main__":
    main()
