import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
W, H, it = 1920, 1080, 100
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
torch.cuda.synchronize()
for graph in (False, True):
    p = hs.make_params(mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=it, term_type=hs.TERM_ITER, use_graph=graph)
    with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
        ctx.set_frames(seeds[0][0], seeds[0][1])
        for _ in range(20): ctx.solve_async(p)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): ctx.solve_async(p)
        ctx.synchronize()
        print("classic one context graph=%d: %.4f ms" % (graph, (time.perf_counter() - t0) / 200 * 1e3))
    for depth in (2, 3):
        with hs.PairPipeline(W, H, depth=depth) as pl:
            def go(n):
                for k in range(n):
                    pl.submit_device(seeds[k & 1][0], seeds[k & 1][1], params=p)
                pl.drain()
            go(30)
            torch.cuda.synchronize(); t0 = time.perf_counter(); go(200); torch.cuda.synchronize()
            print("classic stream depth %d graph=%d: %.4f ms per pair" % (depth, graph, (time.perf_counter() - t0) / 200 * 1e3))
