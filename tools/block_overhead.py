#!/usr/bin/env python3
"""What a timed block of K stream steps costs beyond K x the steady per-pair time (bench.py's bracket: synchronize, K
submissions through hsflow_pipeline_submit_device, drain, synchronize): a + b K fitted over several K.
usage: tools/block_overhead.py [--lanes 2 --depth 6]"""
import argparse
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--depth", type=int, default=6)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iters", type=int, default=100)
args = ap.parse_args()
W, H, it = args.width, args.height, args.iters
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
p = hs.make_params(lam=1.0, max_iter=it, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=float(np.float32(1e-6)), use_graph=True)
with hs.PairPipeline(W, H, depth=args.depth, lanes=args.lanes) as pl:
    n = [0]

    def block(K):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            a, b = seeds[n[0] & 1]
            n[0] += 1
            pl.submit_device(a, b, params=p)
        t1 = time.perf_counter()
        pl.drain()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        return (t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3

    for _ in range(5):
        block(100)
    Ks = [1, 2, 3, 4, 6, 10, 20, 40, 100, 200]
    rows = []
    for K in Ks:
        r = [block(K) for _ in range(9)]
        med = [statistics.median(x[i] for x in r) for i in range(4)]
        rows.append((K, med))
        print("K %4d: block %.4f ms (%.4f per step)   submit loop %.4f   drain %.4f   synchronize %.4f" % (K, med[0], med[0] / K, med[1], med[2], med[3]))
    x = np.array([k for k, _ in rows if k >= 10], dtype=np.float64)
    y = np.array([m[0] for k, m in rows if k >= 10])
    b, a = np.polyfit(x, y, 1)
    print("fit over K >= 10: block = %.4f + %.4f K ms" % (a, b))
