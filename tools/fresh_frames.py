#!/usr/bin/env python3
"""A new resident pair every step through the device-resident pair pipeline (hsflow_pipeline_submit_device), against
back-to-back solves of one pair on one context: what the frame copy, the early-stop check per pair and the slots'
streams cost.   usage: tools/fresh_frames.py [--width W --height H --iters N --steps K]"""
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
ap.add_argument("--steps", type=int, default=400)
args = ap.parse_args()
W, H, it = args.width, args.height, args.iters
eps6 = float(np.float32(1e-6))
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
torch.cuda.synchronize()


def best(fn, n=3):
    fn()
    r = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        r.append((time.perf_counter() - t0) / args.steps * 1e3)
    return min(r)


for tt, name in ((hs.TERM_ITER, "ITER"), (hs.TERM_ITER | hs.TERM_EPS, "ITER|EPS")):
    p = hs.make_params(lam=1.0, max_iter=it, term_type=tt, epsilon=eps6, use_graph=True)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(seeds[0][0], seeds[0][1])

        def same():
            for _ in range(args.steps):
                ctx.solve_async(p)
            ctx.synchronize()

        def fresh_one_ctx():
            for k in range(args.steps):
                ctx.set_frames(seeds[k & 1][0], seeds[k & 1][1])
                ctx.solve_async(p)
            ctx.synchronize()
        print("%dx%d/%d %-8s one context, same pair every step: %.4f ms;  new pair every step (set_frames settles the owed check): %.4f ms"
              % (W, H, it, name, best(same), best(fresh_one_ctx)), flush=True)
    for depth, lanes in ((1, 1), (2, 2), (3, 3), (4, 4), (3, 2), (4, 2), (6, 2), (8, 2), (6, 3), (6, 1)):
        with hs.PairPipeline(W, H, depth=depth, lanes=lanes) as pl:
            def fresh():
                for k in range(args.steps):
                    pl.submit_device(seeds[k & 1][0], seeds[k & 1][1], params=p)
                pl.drain()
            print("   pipeline of %d slots on %d stream(s), new pair every step: %.4f ms" % (depth, lanes, best(fresh)), flush=True)
