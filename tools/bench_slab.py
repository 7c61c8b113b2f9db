#!/usr/bin/env python3
"""BASELINE config C5: ONE large frame split into row slabs over N GPUs, halo rows exchanged with
RCCL send/recv (opticalflowhs_amd/slab.py).  One process per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         tools/bench_slab.py --size 16384 --iters 500 --halo 16

HSFLOW_BENCH_BACKEND=gloo rehearses the same code on a box whose ranks share one GPU (halo rows are
then staged through the host; such a run says nothing about scaling)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--halo", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--overlap", action="store_true", help="two sub-slabs per rank: the exchange of one runs under the sweeps of the other")
    ap.add_argument("--check", action="store_true", help="compare the owned rows with a whole-frame solve on this rank's GPU")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    W = args.width or args.size
    H = args.height or args.size
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    backend = os.environ.get("HSFLOW_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    import opticalflowhs_amd as hs
    from opticalflowhs_amd import slab, synth
    if args.overlap:
        # two sub-slabs per rank, each context on its own torch stream: the RCCL exchange of one sub-slab is
        # in flight while the other one is swept
        s = slab.OverlappedSlabSolver(dist, rank, world, W, H, args.halo,
                                      lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, local, torch_stream=torch.cuda.Stream(device=local), first_row=r0),
                                      stage_on_host=(backend != "nccl"))
    else:
        # the context shares torch's current stream: sweeps, halo copies and RCCL are ordered on the device,
        # the host only enqueues
        tstream = torch.cuda.Stream(device=local)
        torch.cuda.set_stream(tstream)
        s = slab.SlabSolver(dist, rank, world, W, H, args.halo,
                            lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, local, stream=tstream.cuda_stream, first_row=r0),
                            stage_on_host=(backend != "nccl"))
    r0, r1 = s.local_frame_rows()
    A, B = synth.translating_pair(W, H, seed=3, row0=r0, rows=r1 - r0)  # each rank generates only its rows
    s.set_frames(A, B)
    s.solve(1.0, min(args.iters, 2 * args.halo))  # warm-up (plans, allocations)
    times = []
    for _ in range(args.steps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ex = s.solve(1.0, args.iters)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        times.append(time.perf_counter() - t0)
    t = min(times)
    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "Mpixel*iterations/sec, one frame in row slabs", "value": W * H * args.iters / t / 1e6,
                          "unit": "Mpix*iter/s", "n_gpus": world, "seconds": t, "width": W, "height": H, "iters": args.iters,
                          "halo_rows": args.halo, "overlapped": bool(args.overlap), "exchanges": n_ex, "bytes_per_exchange_per_boundary": 2 * 2 * args.halo * W * 4,
                          "backend": backend}))
    if args.check:  # the slab result must be the single-GPU result, bit for bit
        Af, Bf = synth.translating_pair(W, H, seed=3)
        with hs.HSFlow(W, H, own_stream=True) as full:
            full.set_frames(Af, Bf)
            full.solve(lam=1.0, max_iter=args.iters, term_type=hs.TERM_ITER)
            uf, vf = full.flow()
        uo, vo = s.owned_flow()
        ok = bool(np.array_equal(uo, uf[s.lo:s.hi]) and np.array_equal(vo, vf[s.lo:s.hi]))
        print("rank %d rows [%d, %d): %s" % (rank, s.lo, s.hi, "identical to the whole-frame solve" if ok else "MISMATCH"), flush=True)
        if not ok:
            sys.exit(3)
    s.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
