#!/usr/bin/env python3
"""AUTO against the one-sweep-per-launch kernel on awkward shapes: the planner must never be much worse."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
for (W, H) in ((1, 1), (7, 5), (64, 64), (5000, 3), (3, 5000), (16384, 64), (64, 16384), (257, 4099), (8192, 8192), (1921, 1081)):
    A, B = synth.random_pair(W, H, seed=4)
    row = []
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for k, name in ((hs.KERNEL_AUTO, "auto"), (hs.KERNEL_SIMPLE, "simple")):
            p = ctx.make_params(lam=1.0, max_iter=50, term_type=hs.TERM_ITER, kernel=k, use_graph=True)
            for _ in range(2):
                ctx.solve_async(p)
            ctx.synchronize()
            n = 5 if W * H > 4e6 else 20
            t0 = time.perf_counter()
            for _ in range(n):
                ctx.solve_async(p)
            ctx.synchronize()
            row.append((time.perf_counter() - t0) / n * 1e3)
            i = ctx.info()
            if k == hs.KERNEL_AUTO:
                plan = "k%d T%d R%d thr%d tiles%d" % (i["kernel"], i["fuse_steps"], i["groups_per_thread"], i["threads"], i["tiles"])
    print("%6dx%-6d auto %.4f ms  simple %.4f ms  ratio %.2f  (%s)" % (W, H, row[0], row[1], row[0] / row[1], plan), flush=True)
