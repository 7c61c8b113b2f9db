#!/usr/bin/env python3
"""Wall time of synchronous solves for a few representative (size, iterations, criteria) cases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
for (W, H, it) in ((424, 240, 10), (424, 240, 100), (600, 480, 10), (1920, 1080, 10), (1920, 1080, 100), (3840, 2160, 10)):
    A, B = synth.translating_pair(W, H, seed=3)
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for tt, name in ((hs.TERM_ITER, "ITER"), (hs.TERM_ITER | hs.TERM_EPS, "ITER|EPS")):
            for graph in (False, True):
                p = ctx.make_params(lam=0.1, max_iter=it, epsilon=float(np.float32(1e-6)), term_type=tt, use_graph=graph)
                for _ in range(5):
                    ctx.solve(p)
                t0 = time.perf_counter()
                n = 100
                for _ in range(n):
                    ctx.solve(p)
                dt = (time.perf_counter() - t0) / n * 1e6
                i = ctx.info()
                print("%4dx%-4d it %3d %-8s graph %d: %7.1f us  (T %d, R %d, launches %d)" % (W, H, it, name, graph, dt, i["fuse_steps"], i["groups_per_thread"], i["jacobi_launches"]))
