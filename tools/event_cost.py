#!/usr/bin/env python3
"""What an event record behind every solve costs a two-stream stream of solves (the queues then carry a signal packet
between the solves' kernels).  usage: tools/event_cost.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

W, H, it, steps = 1920, 1080, 100, 400
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
torch.cuda.synchronize()


def run(name, nctx, nstreams, events, wait_old):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    ctxs = [hs.HSFlow(W, H, 1, stream=streams[k % nstreams].cuda_stream) for k in range(nctx)]
    evs = [[torch.cuda.Event() for _ in range(events)] for _ in range(nctx)]
    p = hs.make_params(lam=1.0, max_iter=it, term_type=hs.TERM_ITER, use_graph=True)

    def go(k):
        for j in range(k):
            c = j % nctx
            if wait_old and j >= nctx:
                for e in evs[c]:
                    e.synchronize()
            ctxs[c].set_frames(seeds[j & 1][0], seeds[j & 1][1])
            ctxs[c].solve_async(p)
            for e in evs[c]:
                e.record(streams[c % nstreams])
        torch.cuda.synchronize()
    go(60)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(steps)
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print("%-72s %.4f ms per pair" % (name, best), flush=True)
    for c in ctxs:
        c.close()


run("2 contexts / 2 streams, nothing between the solves", 2, 2, 0, False)
run("2 contexts / 2 streams, one event record behind every solve", 2, 2, 1, False)
run("2 contexts / 2 streams, two event records behind every solve", 2, 2, 2, False)
run("8 contexts / 2 streams, nothing between the solves", 8, 2, 0, False)
run("8 contexts / 2 streams, one event, host waits for the slot's old event", 8, 2, 1, True)
run("8 contexts / 2 streams, two events, host waits for the slot's old events", 8, 2, 2, True)
run("2 contexts / 2 streams, one event, host waits for the slot's old event", 2, 2, 1, True)
