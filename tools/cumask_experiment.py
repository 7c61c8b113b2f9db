#!/usr/bin/env python3
"""Experiment: two slots whose streams are confined to DISJOINT halves of the chip (hipExtStreamCreateWithCUMask), so that
the two solves run truly side by side -- one's load phase (memory) under the other's sweeps (VALU) -- instead of taking
turns launch by launch on the whole chip.   usage: tools/cumask_experiment.py [--steps K]"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iters", type=int, default=100)
args = ap.parse_args()
W, H, it = args.width, args.height, args.iters
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
seeds = []
for sd in (1, 2):
    A, B = synth.translating_pair(W, H, seed=sd)
    seeds.append((torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()))
torch.cuda.synchronize()
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """bits: iterable of CU indices (0 .. ncu-1) the stream may use."""
    words = (ncu + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    st = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), m)
    if st:
        raise RuntimeError("hipExtStreamCreateWithCUMask: %d" % st)
    return s


def run(name, masks, term):
    streams = [masked_stream(m) if m is not None else None for m in masks]
    ctxs = [hs.HSFlow(W, H, 1, stream=s.value) if s is not None else hs.HSFlow(W, H, 1, own_stream=True) for s in streams]
    p = hs.make_params(lam=1.0, max_iter=it, term_type=term, epsilon=float(np.float32(1e-6)), use_graph=True)
    n = len(ctxs)

    def go(k):
        for j in range(k):
            c = ctxs[j % n]
            c.set_frames(seeds[j & 1][0], seeds[j & 1][1])   # (ITER|EPS: settles that slot's owed check first)
            c.solve_async(p)
        for c in ctxs:
            c.synchronize()
    go(60)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(args.steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
    print("%-64s %.4f ms per pair" % (name, best), flush=True)
    for c in ctxs:
        c.close()
    for s in streams:
        if s is not None:
            hip.hipStreamDestroy(s)


for term, tn in ((hs.TERM_ITER, "ITER"), (hs.TERM_ITER | hs.TERM_EPS, "ITER|EPS")):
    run("%s two slots, whole chip each (the pipeline today)" % tn, [None, None], term)
    half = ncu // 2
    run("%s two slots, CUs [0,%d) and [%d,%d)" % (tn, half, half, ncu), [range(0, half), range(half, ncu)], term)
    run("%s two slots, even and odd CUs" % tn, [range(0, ncu, 2), range(1, ncu, 2)], term)
    run("%s two slots, alternating groups of 32" % tn, [[b for b in range(ncu) if (b // 32) % 2 == 0], [b for b in range(ncu) if (b // 32) % 2 == 1]], term)
    q = ncu // 4
    run("%s four slots, a quarter of the CUs each" % tn, [range(k * q, (k + 1) * q) for k in range(4)], term)
    run("%s three slots, whole chip" % tn, [None, None, None], term)
