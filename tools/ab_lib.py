#!/usr/bin/env python3
"""A/B of one library build (HSFLOW_LIB_PATH) at 1080p / 100, ITER: one context back to back, and the two-slot stream.
usage: HSFLOW_LIB_PATH=tools/bin/libhsflow_X.so python tools/ab_lib.py [label [eps]]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

W, H, it = 1920, 1080, 100
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("HSFLOW_LIB_PATH", "shipped"))
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
torch.cuda.synchronize()
if "eps" in sys.argv[2:]:  # the witness kernels: ITER|EPS with an epsilon no sweep undercuts
    p = hs.make_params(lam=1.0, max_iter=it, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=float(np.float32(1e-6)), kernel=hs.KERNEL_STRIP, use_graph=True)
    label += " ITER|EPS"
else:
    p = hs.make_params(lam=1.0, max_iter=it, term_type=hs.TERM_ITER, kernel=hs.KERNEL_STRIP, use_graph=True)
with hs.HSFlow(W, H, 1, own_stream=True) as ctx:
    ctx.set_frames(seeds[0][0], seeds[0][1])
    for _ in range(100):
        ctx.solve_async(p)
    ctx.synchronize()
    one = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(300):
            ctx.solve_async(p)
        ctx.synchronize()
        one.append((time.perf_counter() - t0) / 300 * 1e3)
    u0, v0 = ctx.flow()
with hs.PairPipeline(W, H, depth=2) as pl:
    def go(n):
        for k in range(n):
            pl.submit_device(seeds[k & 1][0], seeds[k & 1][1], params=p)
        pl.drain()
    go(60)
    st = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(300)
        torch.cuda.synchronize()
        st.append((time.perf_counter() - t0) / 300 * 1e3)
print("%-28s one context %.4f ms (min of 5; median %.4f)   stream %.4f ms (median %.4f)   checksum %.6f"
      % (label, min(one), sorted(one)[2], min(st), sorted(st)[2], float(np.abs(u0).sum() + np.abs(v0).sum())))
