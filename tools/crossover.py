#!/usr/bin/env python3
"""Strip vs fold vs LDS-tile kernel (each with its own planner) over frame sizes: where AUTO should switch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (W, H) in ((160, 120), (320, 240), (424, 240), (640, 360), (600, 480), (640, 480), (800, 600), (1024, 768), (1280, 720), (1600, 900), (1920, 1080)):
    A, B = synth.translating_pair(W, H, seed=2)
    row = []
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        for k, name in ((hs.KERNEL_STRIP, "strip"), (hs.KERNEL_FOLD, "fold"), (hs.KERNEL_FUSED, "fused")):
            p = ctx.make_params(lam=1.0, max_iter=iters, term_type=hs.TERM_ITER, kernel=k, use_graph=True)
            for _ in range(5):
                ctx.solve_async(p)
            ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(30):
                    ctx.solve_async(p)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
            row.append("%s %.4f" % (name, best))
    print("%5dx%-5d %8d px: %s" % (W, H, W * H, "  ".join(row)), flush=True)
