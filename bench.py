#!/usr/bin/env python3
"""bench.py -- Horn-Schunck hot path on MI355X: Mpixel*iterations/sec.

One "step" = one pass of the hot path (derivative kernel + `iters` Jacobi sweeps) over one batch
of `pairs` synthetic image pairs per GPU, frames already resident in HBM, result left in HBM.
Default workload = BASELINE.json configs[1]: one 1920x1080 translating-texture pair (seed 1),
lambda 1, 100 iterations, fp32, 1 GPU.  With --gpus N every rank runs the same per-GPU batch on its
own pairs (independent pairs shard with no collective: "scaling": "weak"); the only communication
is the barrier + max-over-ranks of the elapsed time.

Launch: python bench.py [--gpus 1]   or, for N > 1,
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
ALG_BYTES_PER_PX_ITER = 28.0    # read Ix,Iy,It,u,v + write u',v' as fp32 planes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pairs", type=int, default=1, help="pairs per GPU per step")
    ap.add_argument("--lam", type=float, default=1.0)
    ap.add_argument("--kernel", choices=["auto", "simple", "fused", "strip"], default="auto")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--fuse-steps", type=int, default=0)
    ap.add_argument("--tile-w", type=int, default=0)
    ap.add_argument("--tile-h", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--iter-eps", action="store_true",
                    help="time the reference's own call form ITER|EPS (eps 1e-6) instead of ITER only (synchronous solves)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work of the single-thread cpu_baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=0, help="iterations of the CPU sample (0: same as --iters)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run\n" % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs a GPU (the product path has no CPU fallback)\n")
        sys.exit(2)
    # one rank per GPU; HSFLOW_BENCH_BACKEND=gloo lets a 1-GPU box rehearse the N > 1 code path
    # (ranks then share the card, so that run says nothing about scaling)
    backend = os.environ.get("HSFLOW_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        sys.stderr.write("bench.py: %d ranks but %d GPUs\n" % (world, ndev))
        sys.exit(2)
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import opticalflowhs_amd as hs
    from opticalflowhs_amd import synth

    W, H, iters, pairs = args.width, args.height, args.iters, args.pairs
    kernel = {"auto": hs.KERNEL_AUTO, "simple": hs.KERNEL_SIMPLE, "fused": hs.KERNEL_FUSED, "strip": hs.KERNEL_STRIP}[args.kernel]
    # all work goes to one non-default stream (the legacy default stream cannot be graph-captured)
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    ctx = hs.HSFlow(W, H, pairs, device=local_rank, stream=stream)
    frames = []
    for i in range(pairs):
        # config C2 uses seed 1; a batch (config C4) uses seed 1000 + global pair index
        seed = 1 if (pairs == 1 and world == 1) else 1000 + rank * pairs + i
        A, B = synth.translating_pair(W, H, seed=seed)
        if i == 0:
            frames = [A, B]
        ctx.set_frames(A, B, pair=i)

    p = ctx.make_params(lam=args.lam, max_iter=iters, term_type=hs.TERM_ITER, kernel=kernel,
                        fuse_steps=args.fuse_steps, tile_w=args.tile_w, tile_h=args.tile_h,
                        threads=args.threads, strip_rows=args.strip_rows, use_graph=not args.no_graph)

    def barrier():
        if world > 1:
            dist.barrier()

    if args.iter_eps:  # the early-stop check needs one read-back per solve: synchronous solves
        p = ctx.make_params(lam=args.lam, max_iter=iters, term_type=hs.TERM_ITER | hs.TERM_EPS,
                            epsilon=float(np.float32(1e-6)), kernel=kernel, fuse_steps=args.fuse_steps,
                            tile_w=args.tile_w, tile_h=args.tile_h, threads=args.threads, strip_rows=args.strip_rows,
                            use_graph=not args.no_graph)
        step = lambda: ctx.solve(p)
    else:
        step = lambda: ctx.solve_async(p)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    info = ctx.info()

    # Per-kernel durations with HIP events on the launch stream: same K steps again, eager launches
    # bracketed by hipEventRecord inside the C ABI (params.profile).  Events between launches would
    # perturb the timed region above, so they run right after it, same process, same buffers.
    pp = ctx.make_params(lam=args.lam, max_iter=iters, term_type=hs.TERM_ITER, kernel=kernel,
                         fuse_steps=args.fuse_steps, tile_w=args.tile_w, tile_h=args.tile_h,
                         threads=args.threads, strip_rows=args.strip_rows, profile=True)
    jac_ms = der_ms = 0.0
    launches = 0
    nprof = max(1, min(args.steps, 50))
    for _ in range(nprof):
        pi = ctx.solve(pp)
        jac_ms += pi["jacobi_ms"]
        der_ms += pi["deriv_ms"]
        launches += pi["jacobi_launches"]

    # the same workload with the termination criteria the reference itself passes (ITER|EPS, eps 1e-6,
    # OpticalFlowOpenCV.cpp:29): synchronous solves, reported beside the headline, not instead of it
    eps_line = None
    if not args.iter_eps and world == 1:
        pe = ctx.make_params(lam=args.lam, max_iter=iters, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=float(np.float32(1e-6)),
                             kernel=kernel, fuse_steps=args.fuse_steps, tile_w=args.tile_w, tile_h=args.tile_h, threads=args.threads,
                             strip_rows=args.strip_rows, use_graph=not args.no_graph)
        for _ in range(3):
            ctx.solve(pe)
        ne = max(5, min(args.steps, 30))
        te = time.perf_counter()
        for _ in range(ne):
            ie = ctx.solve(pe)
        te = (time.perf_counter() - te) / ne
        eps_line = {"criteria": "ITER|EPS, eps 1e-6 (synchronous solves)", "ms_per_step": te * 1e3,
                    "value": W * H * pairs * iters / te / 1e6, "iterations_done": ie["iterations_done"], "eps_rerun": ie["eps_rerun"]}

    KNAME = {hs.KERNEL_SIMPLE: "simple", hs.KERNEL_FUSED: "fused", hs.KERNEL_STRIP: "strip", hs.KERNEL_FOLD: "fold"}
    px = W * H * pairs
    ms_per_step = elapsed / args.steps * 1e3
    value = world * px * iters * args.steps / elapsed / 1e6
    avg_launch_ms = jac_ms / max(launches, 1)
    sweeps_per_launch = iters * nprof / max(launches, 1)
    alg_bytes_per_launch = ALG_BYTES_PER_PX_ITER * px * sweeps_per_launch
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9

    out = {
        "metric": "Mpixel*iterations/sec (Horn-Schunck: derivative pass + Jacobi u/v sweeps, frames resident in HBM)",
        "value": value, "unit": "Mpix*iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d translating-texture pair(s), %d pair(s) per GPU per step, lambda %g, %d Jacobi iterations, ITER termination"
                               % (W, H, pairs, args.lam, iters),
                   "width": W, "height": H, "iters": iters, "pairs_per_gpu": pairs, "lambda": args.lam,
                   "kernel": KNAME[info["kernel"]],
                   "fuse_steps": info["fuse_steps"], "tile": [info["tile_w"], info["tile_h"]],
                   "threads": info["threads"], "rows_per_lane_or_groups": info["groups_per_thread"], "tiles_per_launch": info["tiles"], "lds_bytes": info["lds_bytes"],
                   "hipgraph": not args.no_graph,
                   "termination": "ITER|EPS (eps 1e-6)" if args.iter_eps else "ITER", "sharding": "independent pairs per rank, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_jacobi_" + KNAME[info["kernel"]],
                     "avg_launch_us": avg_launch_ms * 1e3, "sweeps_per_launch": sweeps_per_launch,
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "note": "achieved = 28 B/pixel/sweep x pixels x sweeps per launch / mean launch time (HIP events); "
                             "the kernel runs several sweeps per launch out of registers / LDS, so this can exceed what HBM moves"},
        "kernel_ms_per_step": {"deriv": der_ms / nprof, "jacobi": jac_ms / nprof, "launches": launches / nprof},
    }
    if eps_line:
        out["reference_call_criteria"] = eps_line

    # HBM bytes per launch from the committed PMC summaries (tools/collect_profiles.py): whichever one was
    # taken on this workload with this launch depth
    import glob
    for traffic_file in [os.path.join(ROOT, "profiles", "traffic_latest.json")] + sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            tr = json.load(open(traffic_file))
        except (ValueError, OSError):
            continue
        if tr.get("width") == W and tr.get("height") == H and tr.get("fuse_steps") == info["fuse_steps"] and tr.get("hbm_bytes_per_launch"):
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr.get("source")
            out["roofline"]["traffic_GBps"] = tr["hbm_bytes_per_launch"] / (avg_launch_ms * 1e-3) / 1e9  # what HBM physically moves
            break

    # the roofline fraction also against what a plain device-to-device copy reaches on this box
    # (1 GiB read + 1 GiB written per copy, torch's copy kernel, HIP events on torch's stream)
    if rank == 0:
        try:
            n = 1 << 28  # floats
            src = torch.empty(n, dtype=torch.float32, device="cuda")
            dst = torch.empty_like(src)
            src.fill_(1.0)
            for _ in range(3):
                dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 10
            for _ in range(reps):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 2.0 * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
            out["roofline"]["measured_copy_GBps"] = copy_gbps
            out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbps
            del src, dst
        except RuntimeError:
            pass

    if rank == 0 and world == 1 and not args.skip_cpu:
        from oracle import hs_oracle  # cpu_baseline leg only: the oracle timed as the CPU port
        hs_oracle.build()
        A, B = frames
        cit = args.cpu_iters or iters

        def timed(threads, budget_s):
            """Whole solves of the same pair until about budget_s of CPU work is done (at least one); the first
            one also pays the page faults of the record planes, like a one-off call of the original would."""
            n, t0 = 0, time.perf_counter()
            while True:
                hs_oracle.calc_optical_flow_hs(A, B, args.lam, cit, term_type=hs_oracle.TERMCRIT_ITER, threads=threads)
                n += 1
                el = time.perf_counter() - t0
                if el >= budget_s or n >= 64:
                    return n, el

        n1, t1 = timed(1, args.cpu_seconds)
        out["cpu_baseline"] = {"value": W * H * cit * n1 / t1 / 1e6, "unit": "Mpix*iter/s", "cores": 1, "kind": "port",
                               "sample": "%d solves of one %dx%d pair, %d iterations each, single thread (the original is scalar "
                                         "single-threaded code), %.1f s" % (n1, W, H, cit, t1)}
        nth = hs_oracle.num_threads()
        n2, t2 = timed(0, 0.3 * args.cpu_seconds)
        out["cpu_baseline_all_cores"] = {"value": W * H * cit * n2 / t2 / 1e6, "unit": "Mpix*iter/s", "cores": nth, "kind": "port",
                                         "sample": "%d solves of the same pair, OpenMP row-parallel form, %.1f s" % (n2, t2)}
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
