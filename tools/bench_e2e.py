#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the pair pipeline: page-locked host frames in, host flow out.
This is NOT bench.py's `value` (that one starts with the frames resident in HBM); DESIGN.md quotes
these numbers beside it.  Prints one JSON line per configuration, appends to gpurun_out/e2e.txt.

usage: python tools/bench_e2e.py [--width 1920 --height 1080 --iters 100 --pairs 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pairs", type=int, default=200)
    ap.add_argument("--depths", default="1,2,3,4")
    ap.add_argument("--frames", default="gray", choices=["gray", "gray_blur", "bgr", "bgr_blur"],
                    help="layout of the host frames (bgr*: 3 bytes / pixel, pre-processing on the device)")
    args = ap.parse_args()

    import torch
    import opticalflowhs_amd as hs
    from opticalflowhs_amd import synth

    W, H = args.width, args.height
    nbuf = 6
    A, B = synth.translating_pair(W, H, seed=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "e2e.txt"), "a")

    def emit(d):
        line = json.dumps(d)
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()

    # raw copy rates of this box, for context
    hp = torch.empty(W * H * 4, dtype=torch.uint8).pin_memory()
    dv = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
    for name, fn in (("h2d", lambda: dv.copy_(hp, non_blocking=True)), ("d2h", lambda: hp.copy_(dv, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        emit({"copy": name, "bytes": W * H * 4, "GB/s": round(W * H * 4 * 50 / (time.perf_counter() - t0) / 1e9, 2)})

    for pinned in (True, False):
        for depth in [int(x) for x in args.depths.split(",")]:
            mk = hs.pinned_empty if pinned else (lambda s, d: np.empty(s, d))
            bufs = []
            for i in range(max(nbuf, depth + 1)):
                fshape = (H, W, 3) if args.frames.startswith("bgr") else (H, W)
                a, b = mk(fshape, np.uint8), mk(fshape, np.uint8)
                a[...] = A[..., None] if len(fshape) == 3 else A
                b[...] = B[..., None] if len(fshape) == 3 else B
                bufs.append((a, b, mk((H, W), np.float32), mk((H, W), np.float32)))
            with hs.PairPipeline(W, H, depth=depth) as pl:
                p = hs.make_params(lam=1.0, max_iter=args.iters, term_type=hs.TERM_ITER, use_graph=True)
                for i in range(2 * len(bufs)):
                    pl.submit(*bufs[i % len(bufs)], params=p, frames=args.frames)
                pl.drain()
                t0 = time.perf_counter()
                for i in range(args.pairs):
                    pl.submit(*bufs[i % len(bufs)], params=p, frames=args.frames)  # a buffer set is reused only after its slot was recycled
                pl.drain()
                dt = time.perf_counter() - t0
            emit({"e2e": "pipeline", "frames": args.frames, "pinned": pinned, "depth": depth, "width": W, "height": H, "iters": args.iters,
                  "pairs": args.pairs, "pairs_per_s": round(args.pairs / dt, 1), "ms_per_pair": round(dt / args.pairs * 1e3, 4),
                  "mpix_iter_per_s": round(W * H * args.iters * args.pairs / dt / 1e6, 0),
                  "pcie_GB_per_s": round(((6 if args.frames.startswith("bgr") else 2) * W * H + 8 * W * H) * args.pairs / dt / 1e9, 2)})
    # resident reference point: same solve, no host traffic
    with hs.HSFlow(W, H, own_stream=True) as ctx:
        ctx.set_frames(A, B)
        p = ctx.make_params(lam=1.0, max_iter=args.iters, term_type=hs.TERM_ITER, use_graph=True)
        for _ in range(10):
            ctx.solve_async(p)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.pairs):
            ctx.solve_async(p)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    emit({"e2e": "resident", "ms_per_pair": round(dt / args.pairs * 1e3, 4),
          "mpix_iter_per_s": round(W * H * args.iters * args.pairs / dt / 1e6, 0)})


if __name__ == "__main__":
    main()
