#!/usr/bin/env python3
"""Quick variant sweep of the classic fused kernel (fuse depth x threads) at one frame size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
W, H, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A, B = synth.translating_pair(W, H, seed=1)
rows = []
with hs.HSFlow(W, H, own_stream=True) as ctx:
    ctx.set_frames(A, B)
    for T in (2, 4, 5, 6, 8, 10, 12, 16, 20):
        for nt in (0, 256, 512, 1024):
            try:
                p = ctx.make_params(mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=iters, term_type=hs.TERM_ITER, kernel=hs.KERNEL_FUSED, fuse_steps=T, threads=nt)
                for _ in range(3):
                    ctx.solve_async(p)
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    ctx.solve_async(p)
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / 10 * 1e3
                i = ctx.info()
                rows.append((dt, T, nt, i["tile_w"], i["tile_h"], i["threads"], i["groups_per_thread"], i["tiles"]))
            except hs.HsflowError as e:
                pass
rows.sort()
for r in rows[:12]:
    print("%.4f ms  T=%d nt=%d tile %dx%d threads %d K %d tiles %d" % r)
