#!/usr/bin/env python3
"""Launch shapes for a STREAM of resident pairs (hsflow_pipeline_submit_device): with two slots' streams in flight the
best shape need not be the one a single solve likes -- workgroups of 8 wavefronts leave room for a second workgroup on
the CU, whose sweeps then run beside the first one's load phase.
usage: tools/sweep_pipeline.py [--width W --height H --iters N --steps K] > profiles/..."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--depths", type=int, nargs="*", default=[2])
ap.add_argument("--T", type=int, nargs="*", default=[6, 8, 10, 12, 14, 16, 20])
ap.add_argument("--R", type=int, nargs="*", default=[3, 4, 5])
ap.add_argument("--NW", type=int, nargs="*", default=[6, 8, 10, 12, 16])
ap.add_argument("--fold", action="store_true")
ap.add_argument("--lanes", type=int, default=0, help="streams the slots share (0: one per slot)")
ap.add_argument("--term", choices=["iter", "itereps"], default="itereps")
args = ap.parse_args()
W, H, it = args.width, args.height, args.iters
eps6 = float(np.float32(1e-6))
tt = hs.TERM_ITER if args.term == "iter" else hs.TERM_ITER | hs.TERM_EPS
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
torch.cuda.synchronize()
print("name,depth,ms_per_pair,mpix_iter_per_s,tiles,threads,rows,T,tile_w,tile_h")


def run(name, depth, **kw):
    p = hs.make_params(lam=1.0, max_iter=it, term_type=tt, epsilon=eps6, use_graph=True, **kw)
    try:
        with hs.PairPipeline(W, H, depth=depth, lanes=args.lanes or None) as pl:
            def go(n):
                for k in range(n):
                    pl.submit_device(seeds[k & 1][0], seeds[k & 1][1], params=p)
                pl.drain()
            go(40)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                go(args.steps)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
            t = pl.submit_device(seeds[0][0], seeds[0][1], params=p)
            i = pl.info(t)
    except hs.HsflowError as e:
        return
    print("%s,%d,%.4f,%.0f,%d,%d,%d,%d,%d,%d" % (name, depth, best, W * H * it / best / 1e3, i["tiles"], i["threads"], i["groups_per_thread"],
                                                 i["fuse_steps"], i["tile_w"], i["tile_h"]), flush=True)


for d in args.depths:
    run("auto", d)
kern = hs.KERNEL_FOLD if args.fold else hs.KERNEL_STRIP
for T in args.T:
    for R in args.R:
        for NW in args.NW:
            rows = NW * R * (2 if args.fold else 1)
            if rows - 2 * T < 8:
                continue
            for d in args.depths:
                run("%s_T%d_R%d_NW%d" % ("F" if args.fold else "S", T, R, NW), d, kernel=kern, fuse_steps=T, strip_rows=R, threads=NW * 64)
