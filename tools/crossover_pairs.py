#!/usr/bin/env python3
"""Strip vs fold for batches of small frames (n pairs per context): which one AUTO should pick."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
for (W, H, N) in ((424, 240, 4), (424, 240, 16), (424, 240, 64), (640, 480, 4), (640, 480, 16), (1280, 720, 4), (1280, 720, 8)):
    row = []
    with hs.HSFlow(W, H, N, own_stream=True) as ctx:
        for i in range(N):
            A, B = synth.translating_pair(W, H, seed=1000 + i)
            ctx.set_frames(A, B, pair=i)
        for k, name in ((hs.KERNEL_STRIP, "strip"), (hs.KERNEL_FOLD, "fold")):
            p = ctx.make_params(lam=1.0, max_iter=100, term_type=hs.TERM_ITER, kernel=k, use_graph=True)
            for _ in range(3):
                ctx.solve_async(p)
            ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(10):
                    ctx.solve_async(p)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
            row.append("%s %.4f ms" % (name, best))
    print("%4dx%-4d x %2d pairs (%8d px): %s" % (W, H, N, W * H * N, "  ".join(row)), flush=True)
