#!/usr/bin/env python3
"""Cost of the derivative pass inside a solve: a solve with and without it (reuse_derivatives), hipGraph
replay.  Run once as is (the pass rides in the first Jacobi launch where it can) and once with
HSFLOW_NO_DERIV_FUSION=1 (separate derivative kernel) to compare the two on the same box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
for (W, H, it) in ((1920, 1080, 100), (1920, 1080, 10), (600, 480, 10), (600, 480, 100), (424, 240, 50), (424, 240, 10), (3840, 2160, 200)):
    A, B = synth.translating_pair(W, H, seed=1)
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        out = []
        for reuse in (False, True):
            p = ctx.make_params(lam=1.0, max_iter=it, term_type=hs.TERM_ITER, use_graph=True, reuse_derivatives=reuse)
            for _ in range(10):
                ctx.solve_async(p)
            ctx.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(100):
                    ctx.solve_async(p)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
            out.append(best)
        print("%dx%d it %d: with derivative pass %.1f us, without %.1f us (%.1f %%)" % (W, H, it, out[0], out[1], 100 * (out[0] - out[1]) / out[0]))
