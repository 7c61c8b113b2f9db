#!/usr/bin/env python3
"""bench.py -- Horn-Schunck hot path on MI355X: Mpixel*iterations/sec.

One "step" = one pass of the hot path (derivative pass + `iters` Jacobi sweeps) over one batch of `pairs`
synthetic image pairs per GPU, frames already resident in HBM, result left in HBM.  Default workload =
BASELINE.json configs[1]: one 1920x1080 translating-texture pair (seed 1), lambda 1, 100 iterations, fp32,
1 GPU.  The solver is called the way the reference calls it -- ITER|EPS with eps 1e-6
(OpticalFlowOpenCV.cpp:29) -- through hsflow_solve_async; the ITER-only form is timed beside it.
With --gpus N every rank runs the same per-GPU batch on its own pairs (independent pairs shard with no
collective: "scaling": "weak"); the only communication of the headline is the barrier + max-over-ranks of
the elapsed time.  For N > 1 two more measurements ride in the same JSON line (extra keys): `c4_pipeline`
(BASELINE config C4: 512 host-resident pairs over the ranks through the pair pipeline, PCIe included) and
`c5_slab` (config C5: one 16384^2 frame in row slabs, halo rows exchanged with RCCL send/recv).

Launch: python bench.py [--gpus N]   (for N > 1 without WORLD_SIZE in the environment this process starts the N
        ranks itself -- fresh child processes under torch.distributed.run, before anything here touches the GPU --
        relays rank 0's JSON line and exits with the children's status), or, as the driver does for N > 1,
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
ALG_BYTES_PER_PX_ITER = 28.0    # read Ix,Iy,It,u,v + write u',v' as fp32 planes
ALG_FLOPS_PER_PX_ITER = 22.0    # SURVEY.md 8d (informational flop count of one update)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table
# VALU-issue roofline (the bound that binds the multi-sweep kernels):
N_SIMD = 256 * 4                # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
SPEC_CLOCK_GHZ = 2.4            # chip table, max clock
SPEC_CYCLES_PER_WAVE64_OP = 2.0  # SIMD-32: a wave64 VALU instruction issues over 2 cycles
MEASURED_CYCLES_PER_WAVE64_OP = 2.3   # profiles/r01_ubench_valu.txt: v_add/fma/mov_f32 at 2-4 wavefronts per SIMD
MEASURED_CLOCK_GHZ = 2.14             # profiles/r01_phase_stamps_1080p.txt: s_memtime / s_memrealtime under this kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps (the first one is `value`; the rest give the spread)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pairs", type=int, default=1, help="pairs per GPU per step")
    ap.add_argument("--lam", type=float, default=1.0)
    ap.add_argument("--kernel", choices=["auto", "simple", "fused", "strip", "fold"], default="auto")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--fuse-steps", type=int, default=0)
    ap.add_argument("--tile-w", type=int, default=0)
    ap.add_argument("--tile-h", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--iter-only", action="store_true", help="headline with ITER termination instead of the reference's ITER|EPS")
    ap.add_argument("--no-side", action="store_true", help="skip the timing of the other termination form (profiling runs)")
    ap.add_argument("--sync-solves", action="store_true", help="hsflow_solve (host waits for every solve) instead of hsflow_solve_async")
    ap.add_argument("--loop", choices=["auto", "stream", "repeat"], default="auto",
                    help="what a step is: `stream` = a NEW resident pair every step through the device-resident pair pipeline (two seed pairs "
                         "alternating; every pair pays its frame copy, derivative pass and its own early-stop check) -- the default for one pair "
                         "per step with the strip / fold kernels; `repeat` = the same resident batch solved again on one context")
    ap.add_argument("--stream-depth", type=int, default=6, help="slots of the pair pipeline in the `stream` loop")
    ap.add_argument("--stream-lanes", type=int, default=2, help="streams those slots are spread over (two solves side by side fill the chip's "
                                                                 "gaps; the further slots keep both streams' queues full)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work of the single-thread cpu_baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=0, help="iterations of the CPU sample (0: same as --iters)")
    # multi-GPU extras (default: on for N > 1, off for N = 1)
    ap.add_argument("--c4", choices=["auto", "on", "off"], default="auto", help="config C4: host-resident pairs through the pair pipeline")
    ap.add_argument("--c4-pairs", type=int, default=512, help="pairs of config C4 over all ranks")
    ap.add_argument("--c4-pool", type=int, default=8, help="distinct synthetic pairs per rank that the C4 submissions cycle through")
    ap.add_argument("--c5", choices=["auto", "on", "off"], default="auto", help="config C5: one frame in row slabs with halo exchange")
    ap.add_argument("--side-timeout", type=float, default=240.0,
                    help="N > 1: seconds the C4 / C5 measurements may take together before the line is printed without them")
    ap.add_argument("--debug-hang", choices=["", "c4", "c5"], default="", help=argparse.SUPPRESS)  # tests: a side measurement that never returns
    ap.add_argument("--c5-size", type=int, default=16384)
    ap.add_argument("--c5-iters", type=int, default=500)
    ap.add_argument("--c5-halo", type=int, default=16)
    ap.add_argument("--c5-check", choices=["on", "off"], default="on", help="owned rows against a rank-local band solve, bit for bit")
    return ap.parse_args()


def gen_rows(synth, W, H, seed, row0, rows, block=512):
    """Rows [row0, row0+rows) of the translating-texture pair, generated in blocks (bounded host memory)."""
    A = np.empty((rows, W), np.uint8)
    B = np.empty((rows, W), np.uint8)
    for r in range(0, rows, block):
        n = min(block, rows - r)
        A[r:r + n], B[r:r + n] = synth.translating_pair(W, H, seed=seed, row0=row0 + r, rows=n)
    return A, B


def run_c4(args, hs, synth, dist, world, rank, local_rank, barrier, reduce_max):
    """BASELINE config C4 with the PCIe legs: --c4-pairs host-resident 1080p pairs sharded over the ranks
    (pair i -> rank i mod world, slab.shard_pairs), each rank streaming its share through the pair pipeline at
    depth 4 (page-locked buffers; upload, solve and download of neighbouring pairs overlap).  The submissions
    cycle through a pool of distinct synthetic pairs (seeds 1000 + global pair index); timing does not depend
    on the pixel values."""
    from opticalflowhs_amd import slab
    W, H, it = 1920, 1080, 100
    mine = slab.shard_pairs(args.c4_pairs, world, rank)
    pool = max(1, min(args.c4_pool, len(mine)))
    ins = []
    for k in range(pool):
        A, B = synth.translating_pair(W, H, seed=1000 + mine[k])
        a, b = hs.pinned_empty((H, W), np.uint8), hs.pinned_empty((H, W), np.uint8)
        a[...], b[...] = A, B
        ins.append((a, b))
    depth = 4
    outs = [(hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)) for _ in range(depth + 1)]
    eps6 = float(np.float32(1e-6))
    p = hs.make_params(lam=1.0, max_iter=it, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=eps6, use_graph=True)
    with hs.PairPipeline(W, H, depth=depth, device=local_rank) as pl:
        for k in range(min(2 * depth, len(mine))):  # warm-up: plans, graphs, first-touch of the buffers
            pl.submit(ins[k % pool][0], ins[k % pool][1], outs[k % (depth + 1)][0], outs[k % (depth + 1)][1], params=p)
        pl.drain()
        barrier()
        t0 = time.perf_counter()
        for k in range(len(mine)):
            pl.submit(ins[k % pool][0], ins[k % pool][1], outs[k % (depth + 1)][0], outs[k % (depth + 1)][1], params=p)
        pl.drain()
        barrier()
        el = reduce_max(time.perf_counter() - t0)
    return {"config": "C4: %d host-resident 1920x1080 pairs, 100 iterations (ITER|EPS, eps 1e-6), pair pipeline depth %d, PCIe inclusive"
                      % (args.c4_pairs, depth),
            "pairs_total": args.c4_pairs, "pairs_this_rank": len(mine), "distinct_pairs_per_rank": pool, "seconds": el,
            "pairs_per_s": args.c4_pairs / el, "value": args.c4_pairs * W * H * it / el / 1e6, "unit": "Mpix*iter/s",
            "collective": "none (barrier for timing only)"}


def run_fresh_frames(args, hs, synth, torch, local_rank, W, H, iters, p_ieps, barrier, reduce_max, world):
    """The reference's camera loop at solver speed (OpticalFlowOpenCV.cpp:91-95: a NEW pair every step, ITER|EPS):
    two different seed pairs resident in HBM, alternated step by step through the device-resident pair pipeline
    (hsflow_pipeline_submit_device: frames copied device to device into the slot, pair k+1 enqueued while pair k's
    early-stop check is still owed; the check is looked at when the slot comes round again).  Every step pays its own
    frame copy, derivative pass and early-stop check -- nothing is carried over between steps."""
    depth = args.stream_depth
    seeds = []
    for sd in (1, 2):
        A, B = synth.translating_pair(W, H, seed=sd)
        seeds.append((torch.from_numpy(A).to("cuda:%d" % local_rank), torch.from_numpy(B).to("cuda:%d" % local_rank)))
    torch.cuda.synchronize()
    with hs.PairPipeline(W, H, depth=depth, device=local_rank, lanes=min(args.stream_lanes, depth)) as pl:
        def run(n):
            for k in range(n):
                a, b = seeds[k & 1]
                pl.submit_device(a, b, params=p_ieps)
            pl.drain()
        run(max(4 * depth, min(args.warmup, 50)))
        blocks = []
        for _ in range(max(1, min(args.blocks, 3))):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            barrier()
            blocks.append(reduce_max(time.perf_counter() - t0) / args.steps * 1e3)
        t_last = pl.submit_device(seeds[0][0], seeds[0][1], params=p_ieps)
        info = pl.info(t_last)
    ms = statistics.median(blocks)
    return {"what": "a different resident pair every step (two seed pairs alternating), ITER|EPS (eps 1e-6), hsflow_pipeline_submit_device "
                    "at depth %d: per step a device-to-device frame copy, the derivative pass, %d sweeps and that pair's own early-stop check" % (depth, iters),
            "ms_per_step": ms, "ms_per_step_blocks": blocks, "value": world * W * H * iters / (ms * 1e-3) / 1e6, "unit": "Mpix*iter/s",
            "iterations_done": info["iterations_done"], "eps_rerun": info["eps_rerun"], "depth": depth}


def read_pgm(path):
    """Binary PGM (P5, maxval 255) -> (H, W) uint8."""
    with open(path, "rb") as f:
        data = f.read()
    tok, pos = [], 0
    while len(tok) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(data[pos:end])
        pos = end
    if tok[0] != b"P5" or int(tok[3]) != 255:
        raise ValueError("not an 8-bit binary PGM")
    w, h = int(tok[1]), int(tok[2])
    return np.frombuffer(data, np.uint8, w * h, pos + 1).reshape(h, w).copy()


def run_reference_default(args, hs, synth, torch, local_rank, with_cpu):
    """The workload the reference's authors ran (main.cpp:4-8,16,20): the city pair, 600x480, lambda 0.1, 100 iterations,
    the CPU route's call -- 3x3 blur of both frames, then cvCalcOpticalFlowHS with ITER|EPS (OpticalFlowOpenCV.cpp:26-30).
    Frames: the committed gray fixtures of the reference's city_1/2.jpg where present, else a synthetic pair of that size.
    Four figures: one solve after the other on one context with the blurred frames resident (`latency`); a stream of
    resident pairs through the pair pipeline (`stream`: the planner's own launch shape, and large tiles picked by hand);
    end to end from page-locked host memory (gray upload, blur on the device, solve, flow download; `end_to_end`); and the
    CPU port (blur + oracle, one thread) beside them."""
    W, H, it, lam = 600, 480, 100, 0.1
    eps6 = float(np.float32(1e-6))
    g = [os.path.join(ROOT, "tests", "golden", "city_%d_gray.pgm" % k) for k in (1, 2)]
    if all(os.path.exists(x) for x in g):
        A, B = read_pgm(g[0]), read_pgm(g[1])
        src = "tests/golden/city_{1,2}_gray.pgm (the reference's city pair as gray planes)"
    else:
        A, B = synth.translating_pair(W, H, seed=4)
        src = "synthetic translating texture, seed 4"
    out = {"what": "600x480, lambda 0.1, 100 iterations, ITER|EPS (eps 1e-6), 3x3 blur first: main.cpp:4-8 / OpticalFlowOpenCV.cpp:26-30", "frames": src,
           "unit": "ms per pair"}
    p = hs.make_params(lam=lam, max_iter=it, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=eps6, use_graph=True)
    p_big = hs.make_params(lam=lam, max_iter=it, term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=eps6, use_graph=True, kernel=hs.KERNEL_STRIP,
                           fuse_steps=20, strip_rows=5, threads=768)
    n = 300
    with hs.HSFlow(W, H, 1, device=local_rank, own_stream=True) as ctx:
        ctx.set_frames_gray_blur(A, B)
        Ab, Bb = ctx.frames()
        for _ in range(30):
            ctx.solve_async(p)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.solve_async(p)
        ctx.synchronize()
        info = ctx.info()
        out["latency"] = {"ms": (time.perf_counter() - t0) / n * 1e3, "kernel": {hs.KERNEL_STRIP: "strip", hs.KERNEL_FOLD: "fold"}.get(info["kernel"]),
                          "fuse_steps": info["fuse_steps"], "tiles": info["tiles"], "iterations_done": info["iterations_done"],
                          "what": "the blurred pair resident, one hsflow_solve_async after the other on one context"}
    dA, dB = torch.from_numpy(Ab).to("cuda:%d" % local_rank), torch.from_numpy(Bb).to("cuda:%d" % local_rank)
    torch.cuda.synchronize()
    for name, pp_, depth in (("stream", p, 8), ("stream_shape_by_hand", p_big, 8)):
        with hs.PairPipeline(W, H, depth=depth, device=local_rank) as pl:
            def go(k):
                for _ in range(k):
                    pl.submit_device(dA, dB, params=pp_)
                pl.drain()
            go(4 * depth)
            t0 = time.perf_counter()
            go(n)
            ms = (time.perf_counter() - t0) / n * 1e3
            i2 = pl.info(pl.submit_device(dA, dB, params=pp_))
        out[name] = {"ms": ms, "depth": depth, "tiles": i2["tiles"], "fuse_steps": i2["fuse_steps"], "rows": i2["groups_per_thread"], "threads": i2["threads"],
                     "what": "resident pairs through hsflow_pipeline_submit_device (frame copy, solve, own early-stop check per pair)"}
    out["stream"]["what"] += "; launch shape: the pipeline's own for small frames at depth >= 3 (few large tiles per solve, several solves side by side)"
    out["stream_shape_by_hand"]["what"] += "; the same shape passed explicitly (strip kernel, 20 sweeps per launch, 5 rows per lane, 768 threads)"
    depth = 4
    with hs.PairPipeline(W, H, depth=depth, device=local_rank) as pl:
        a, b = hs.pinned_empty((H, W), np.uint8), hs.pinned_empty((H, W), np.uint8)
        a[...], b[...] = A, B
        outs = [(hs.pinned_empty((H, W), np.float32), hs.pinned_empty((H, W), np.float32)) for _ in range(depth + 1)]

        def go2(k):
            for j in range(k):
                pl.submit(a, b, outs[j % (depth + 1)][0], outs[j % (depth + 1)][1], params=p, frames="gray_blur")
            pl.drain()
        go2(3 * depth)
        t0 = time.perf_counter()
        go2(n)
        out["end_to_end"] = {"ms": (time.perf_counter() - t0) / n * 1e3, "depth": depth,
                             "what": "page-locked host gray frames in, flow out: upload, 3x3 blur on the device, solve, download (hsflow_pipeline_submit_ex)"}
    if with_cpu:
        from oracle import hs_oracle  # cpu_baseline leg only: the oracle timed as the CPU port
        hs_oracle.build()
        t0 = time.perf_counter()
        reps = 0
        while reps < 1 or time.perf_counter() - t0 < 2.0:
            a2, b2 = hs_oracle.box_blur3(A), hs_oracle.box_blur3(B)
            hs_oracle.calc_optical_flow_hs(a2, b2, lam, it, epsilon=eps6, term_type=hs_oracle.TERMCRIT_ITER | hs_oracle.TERMCRIT_EPS)
            reps += 1
        out["cpu_port"] = {"ms": (time.perf_counter() - t0) / reps * 1e3, "cores": 1, "kind": "port", "what": "blur + oracle, one thread, %d run(s)" % reps}
    return out


def run_c5(args, hs, synth, torch, dist, world, rank, local_rank, backend, barrier, reduce_max):
    """BASELINE config C5: one --c5-size^2 frame (seed 3) in row slabs over the ranks, --c5-iters sweeps in
    chunks of --c5-halo with `halo` rows of u, v swapped between neighbouring ranks after every chunk
    (torch.distributed batch_isend_irecv = RCCL send/recv on the GPU box).  Both drivers are timed: the plain
    one and the overlapped one (two sub-slabs per rank).  Check: each rank re-solves a band = its owned rows
    plus `iters` rows of margin either side in an ordinary one-context solve (exact for the owned rows, a
    border error travels one row per sweep) and compares bit for bit."""
    from opticalflowhs_amd import slab
    S, it, halo = args.c5_size, args.c5_iters, args.c5_halo
    out = {"config": "C5: one %dx%d frame (seed 3), %d iterations, row slabs over %d rank(s), halo %d rows exchanged every %d sweeps"
                     % (S, S, it, world, halo, halo),
           "backend": backend, "unit": "Mpix*iter/s", "bytes_per_exchange_per_boundary": 2 * 2 * halo * S * 4}
    stage = backend != "nccl"
    for name in ("plain", "overlap"):
        if name == "overlap":
            s = slab.OverlappedSlabSolver(dist, rank, world, S, S, halo,
                                          lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, local_rank, torch_stream=torch.cuda.Stream(device=local_rank), first_row=r0),
                                          stage_on_host=stage)
        else:
            ts = torch.cuda.current_stream(local_rank)
            s = slab.SlabSolver(dist, rank, world, S, S, halo,
                                lambda w, h, r0: slab.HSFlowSlabBackend(hs, w, h, local_rank, stream=ts.cuda_stream, first_row=r0), stage_on_host=stage)
        r0, r1 = s.local_frame_rows()
        A, B = gen_rows(synth, S, S, 3, r0, r1 - r0)
        s.set_frames(A, B)
        s.solve(1.0, min(it, 2 * halo))  # warm-up: plans, graphs, communicators
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        n_ex = s.solve(1.0, it)
        torch.cuda.synchronize()
        barrier()
        el = reduce_max(time.perf_counter() - t0)
        out[name] = {"seconds": el, "value": S * S * it / el / 1e6, "exchanges": n_ex}
        if args.c5_check == "on" and name == "plain":
            lo, hi = s.lo, s.hi
            b0, b1 = max(0, lo - it), min(S, hi + it)
            Ab, Bb = (A, B) if (b0, b1) == (r0, r1) else gen_rows(synth, S, S, 3, b0, b1 - b0)
            with hs.HSFlow(S, b1 - b0, 1, device=local_rank, own_stream=True) as band:
                band.set_frames(Ab, Bb)
                band.solve(lam=1.0, max_iter=it, term_type=hs.TERM_ITER)
                ub, vb = band.flow()
            uo, vo = s.owned_flow()
            ok = bool(np.array_equal(uo, ub[lo - b0:hi - b0]) and np.array_equal(vo, vb[lo - b0:hi - b0]))
            del ub, vb, uo, vo, Ab, Bb
            if world > 1:
                t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = bool(t.item() == 1.0)
            out["owned_rows_bit_identical_to_band_solve"] = ok
        s.close()
        del A, B
    out["value"] = max(out["plain"]["value"], out["overlap"]["value"])
    return out


def self_launch(args):
    """--gpus N > 1 started as a plain `python bench.py`: run the N ranks as children of a torch.distributed.run
    child (nothing in THIS process has touched torch or the GPU yet), pass their output through and return their
    exit status.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's hosts support dmabuf IPC only
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
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
    kernel = {"auto": hs.KERNEL_AUTO, "simple": hs.KERNEL_SIMPLE, "fused": hs.KERNEL_FUSED, "strip": hs.KERNEL_STRIP,
              "fold": hs.KERNEL_FOLD}[args.kernel]
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

    eps6 = float(np.float32(1e-6))
    tune = dict(lam=args.lam, max_iter=iters, kernel=kernel, fuse_steps=args.fuse_steps, tile_w=args.tile_w, tile_h=args.tile_h,
                threads=args.threads, strip_rows=args.strip_rows)
    p_iter = ctx.make_params(term_type=hs.TERM_ITER, use_graph=not args.no_graph, **tune)
    p_ieps = ctx.make_params(term_type=hs.TERM_ITER | hs.TERM_EPS, epsilon=eps6, use_graph=not args.no_graph, **tune)

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    def stepper(p):
        return (lambda: ctx.solve(p)) if args.sync_solves else (lambda: ctx.solve_async(p))

    def timed_block(step):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        return reduce_max(time.perf_counter() - t0)

    if args.kernel in ("simple", "fused") and not args.sync_solves:
        # asynchronous ITER|EPS rests on the strip / fold kernels' witness launches; these two kernels are timed ITER only
        args.iter_only, args.no_side = True, True
    head_p, side_p = (p_iter, p_ieps) if args.iter_only else (p_ieps, p_iter)
    head_name, side_name = ("ITER", "ITER|EPS (eps 1e-6)") if args.iter_only else ("ITER|EPS (eps 1e-6)", "ITER")
    # What a step is.  `stream`: the reference's camera loop (OpticalFlowOpenCV.cpp:91-95 -- a NEW pair every step) with the
    # pairs resident in HBM: two seed pairs alternate through hsflow_pipeline_submit_device; pair k+1 is enqueued on the next
    # slot's stream while pair k's early-stop check is still owed, and nothing is carried over from one step to the next.
    # `repeat`: the same resident batch solved again and again on one context (rounds 1-2's loop; kept as `single_context`).
    loop = args.loop
    if loop == "auto":
        loop = "stream" if (pairs == 1 and args.kernel in ("auto", "strip", "fold") and not args.sync_solves and not args.no_graph) else "repeat"
    pl = None
    if loop == "stream":
        if pairs != 1:
            sys.stderr.write("bench.py: --loop stream runs one pair per step\n")
            sys.exit(2)
        s0 = 1 if world == 1 else 1000 + 2 * rank
        dev_pairs = [(torch.from_numpy(frames[0]).to("cuda:%d" % local_rank), torch.from_numpy(frames[1]).to("cuda:%d" % local_rank))]
        A2, B2 = synth.translating_pair(W, H, seed=s0 + 1)
        dev_pairs.append((torch.from_numpy(A2).to("cuda:%d" % local_rank), torch.from_numpy(B2).to("cuda:%d" % local_rank)))
        torch.cuda.synchronize()
        pl = hs.PairPipeline(W, H, depth=args.stream_depth, device=local_rank, lanes=min(args.stream_lanes, args.stream_depth))
        n_sub = [0]

        def stepper(p):  # noqa: F811 -- the stream loop's step: submit the next pair (blocks only while its slot is still busy)
            def step():
                a, b = dev_pairs[n_sub[0] & 1]
                n_sub[0] += 1
                return pl.submit_device(a, b, params=p)
            return step

        def timed_block(step):  # noqa: F811 -- K submissions, then everything in flight drained, inside the bracket
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            pl.drain()
            torch.cuda.synchronize()
            barrier()
            return reduce_max(time.perf_counter() - t0)
    step = stepper(head_p)
    # Warm-up: --warmup untimed steps, and further untimed steps until WARM_SECONDS have passed -- an idle MI355X needs
    # ~30 ms of back-to-back work to reach the clock it then holds (profiles/r02_clock_ramp.txt: 0.196 -> 0.175 ms per
    # step over the first 150 steps), and a timed region that starts inside that ramp measures the ramp.
    WARM_SECONDS = 0.1
    t_w = time.perf_counter()
    n_warm = 0
    while n_warm < args.warmup or time.perf_counter() - t_w < WARM_SECONDS:
        step()
        n_warm += 1
        if n_warm % 16 == 0:
            torch.cuda.synchronize()   # keep the host from running far ahead of the device while watching the clock
    torch.cuda.synchronize()

    # THE timed region of the contract: exactly --steps steps between barrier + synchronize, max over ranks
    elapsed = timed_block(step)
    # (settles the last solve's early-stop check: iterations_done / eps_rerun are final)
    info = pl.info(step()) if pl else ctx.info()
    # spread: the same block again, --blocks - 1 more times
    block_ms = [elapsed / args.steps * 1e3] + [timed_block(step) / args.steps * 1e3 for _ in range(max(0, args.blocks - 1))]
    # the other termination form beside it (same steps, same block structure)
    side_ms, info2 = [float("nan")], info
    if not args.no_side:
        step2 = stepper(side_p)
        for _ in range(min(args.warmup, 5)):
            step2()
        side_ms = [timed_block(step2) / args.steps * 1e3 for _ in range(max(1, min(args.blocks, 3)))]
        info2 = pl.info(step2()) if pl else ctx.info()
    # rounds 1-2's loop beside the stream: the SAME pair solved again and again on one context (an asynchronous ITER|EPS
    # solve that repeats the owed one bit for bit takes its early-stop check over, one check is settled at the end)
    single = None
    if pl is not None:
        pl.drain()
        single = {"what": "the same resident pair solved again and again on ONE context through hsflow_solve_async (one stream; an ITER|EPS "
                          "solve that repeats the owed one takes its early-stop check over -- no check settled inside the timed region)"}
        if not args.no_side:
            for nm, pp_ in (("ITER|EPS (eps 1e-6)", p_ieps), ("ITER", p_iter)):
                st_ = (lambda q: (lambda: ctx.solve_async(q)))(pp_)
                for _ in range(20):
                    st_()
                ms_ = []
                for _ in range(max(1, min(args.blocks, 3))):
                    barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        st_()
                    torch.cuda.synchronize()
                    barrier()
                    ms_.append(reduce_max(time.perf_counter() - t0) / args.steps * 1e3)
                single[nm] = {"ms_per_step": statistics.median(ms_), "ms_per_step_blocks": ms_,
                              "value": world * W * H * iters / (statistics.median(ms_) * 1e-3) / 1e6}
            ctx.synchronize()

    # Per-kernel durations with HIP events on the launch stream: eager launches of the same solve bracketed by
    # hipEventRecord inside the C ABI (params.profile).  Events between graph nodes would perturb the timed
    # region above, so they run right after it, same process, same buffers.  (The derivative pass is a kernel
    # of its own here; in the timed region it rides in the first Jacobi launch.)
    pp = ctx.make_params(term_type=hs.TERM_ITER, profile=True, **tune)
    jac_ms = der_ms = 0.0
    launches = 0
    nprof = max(1, min(args.steps, 50))
    for _ in range(nprof):
        pi = ctx.solve(pp)
        jac_ms += pi["jacobi_ms"]
        der_ms += pi["deriv_ms"]
        launches += pi["jacobi_launches"]

    KNAME = {hs.KERNEL_SIMPLE: "simple", hs.KERNEL_FUSED: "fused", hs.KERNEL_STRIP: "strip", hs.KERNEL_FOLD: "fold"}
    px = W * H * pairs
    ms_per_step = elapsed / args.steps * 1e3
    value = world * px * iters * args.steps / elapsed / 1e6
    avg_launch_ms = jac_ms / max(launches, 1)
    sweeps_per_launch = iters * nprof / max(launches, 1)
    jac_us_per_step = jac_ms / nprof * 1e3

    # --- roofline of the dominant kernel (the Jacobi launch) ------------------------------------------------
    # VALU issue is what binds the multi-sweep kernels: several sweeps run per launch out of registers, so the
    # 28 B/pixel/sweep of SURVEY.md 8d are not what HBM moves (kept below as hbm_algorithmic; it exceeds the HBM
    # peak and is no bound).  op slots = wave64 VALU lane-operations the update needs per pixel and sweep as the
    # kernels compute it (opticalflowhs_amd.OP_SLOTS_PER_PIXEL_SWEEP: 9 with the shared cross sums; the straightforward
    # form costs 11, reported beside it as frac_at_11_op_slots); peak = all SIMDs issuing one such operation per
    # MEASURED_CYCLES_PER_WAVE64_OP cycles at the clock the chip holds under this kernel.
    op_slots = hs.OP_SLOTS_PER_PIXEL_SWEEP
    lane_ops_per_launch = op_slots * px * sweeps_per_launch
    peak_measured = N_SIMD * 64.0 / MEASURED_CYCLES_PER_WAVE64_OP * MEASURED_CLOCK_GHZ * 1e9 / 1e12   # Tlane-op/s
    peak_spec = N_SIMD * 64.0 / SPEC_CYCLES_PER_WAVE64_OP * SPEC_CLOCK_GHZ * 1e9 / 1e12
    achieved_valu = lane_ops_per_launch / (avg_launch_ms * 1e-3) / 1e12
    ideal_us_per_step = op_slots * px * iters / (peak_measured * 1e12) * 1e6
    alg_bytes_per_launch = ALG_BYTES_PER_PX_ITER * px * sweeps_per_launch
    alg_gbps = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    multi = info["kernel"] != hs.KERNEL_SIMPLE
    roof = {
        "bound": "valu" if multi else "hbm",
        "kernel": "k_jacobi_" + KNAME[info["kernel"]],
        "avg_launch_us": avg_launch_ms * 1e3, "sweeps_per_launch": sweeps_per_launch,
        "traffic": None, "traffic_measured_in_run": False,
        "valu_issue": {
            "op_slots_per_pixel_sweep": op_slots, "lane_ops_per_launch": lane_ops_per_launch,
            "achieved_Tlaneops": achieved_valu,
            "peak_Tlaneops_measured_issue": peak_measured, "peak_Tlaneops_spec": peak_spec,
            "frac_of_measured_issue": achieved_valu / peak_measured, "frac_of_spec": achieved_valu / peak_spec,
            "ideal_us_per_step": ideal_us_per_step, "jacobi_kernel_us_per_step": jac_us_per_step,
            "frac_at_11_op_slots": hs.OP_SLOTS_PER_PIXEL_SWEEP_STRAIGHTFORWARD / float(op_slots) * achieved_valu / peak_measured,
            "constants": {"simds": N_SIMD, "cycles_per_wave64_op_measured": MEASURED_CYCLES_PER_WAVE64_OP,
                          "clock_ghz_measured": MEASURED_CLOCK_GHZ, "cycles_per_wave64_op_spec": SPEC_CYCLES_PER_WAVE64_OP,
                          "clock_ghz_spec": SPEC_CLOCK_GHZ,
                          "source": "profiles/r01_ubench_valu.txt (issue cost), profiles/r01_phase_stamps_1080p.txt (clock), "
                                    "MI355X_MICROARCH.md (SIMD-32, 256 CUs x 4 SIMDs, 2.4 GHz)"},
            "note": "ideal_us_per_step = op_slots x pixels x sweeps / measured-issue peak; frac = ideal / measured Jacobi kernel time"},
        "fp32_vector": {"flops_per_pixel_sweep": ALG_FLOPS_PER_PX_ITER,
                        "achieved_TFLOPs": ALG_FLOPS_PER_PX_ITER * px * sweeps_per_launch / (avg_launch_ms * 1e-3) / 1e12,
                        "peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS},
        "hbm_algorithmic": {"bytes_per_pixel_sweep": ALG_BYTES_PER_PX_ITER, "bytes_per_launch": alg_bytes_per_launch, "GBps": alg_gbps,
                            "ratio_to_hbm_peak": alg_gbps / HBM_PEAK_GBS,
                            "note": "SURVEY.md 8d accounting (28 B/pixel/sweep x pixels x sweeps per launch / launch time); with several "
                                    "sweeps per launch this is not what HBM moves and may exceed the peak -- not a bound"},
    }
    roof["fp32_vector"]["frac"] = roof["fp32_vector"]["achieved_TFLOPs"] / FP32_VECTOR_PEAK_TFLOPS
    if multi:
        roof.update({"achieved": achieved_valu, "peak": peak_measured, "unit": "Tlane-op/s", "frac": achieved_valu / peak_measured})
        # `frac` prices the kernel alone on the chip (HIP events around eager launches on one stream, what rocprofv3 sees too);
        # in the stream loop two slots' launches overlap (one pair's load phase and launch gaps under the other's sweeps), so
        # the chip gets through a step in less than the sum of its kernels' own durations:
        roof["valu_issue"]["frac_at_step_rate"] = ideal_us_per_step / (ms_per_step * 1e3)
    else:
        roof.update({"achieved": alg_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbps / HBM_PEAK_GBS})

    # HBM bytes per launch from a committed PMC summary (tools/collect_profiles.py) taken on this very launch shape:
    # kernel, sweeps per launch, rows per lane, workgroup size, frame size.  Not measured in this run (PMC needs
    # rocprofv3); null when no summary matches.
    for traffic_file in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            tr = json.load(open(traffic_file))
        except (ValueError, OSError):
            continue
        want = {"kernel": roof["kernel"], "width": W, "height": H, "pairs": pairs, "fuse_steps": info["fuse_steps"],
                "rows_per_lane_or_groups": info["groups_per_thread"], "threads": info["threads"]}
        if all(tr.get(k) == v for k, v in want.items()) and tr.get("hbm_bytes_per_launch"):
            roof["traffic"] = tr["hbm_bytes_per_launch"]
            roof["hbm_traffic"] = {"bytes_per_launch": tr["hbm_bytes_per_launch"], "measured_in_run": False,
                                   "source": tr.get("source"), "file": os.path.relpath(traffic_file, ROOT), "tag": tr.get("tag"),
                                   "GBps": tr["hbm_bytes_per_launch"] / (avg_launch_ms * 1e-3) / 1e9,
                                   "frac_of_hbm_peak": tr["hbm_bytes_per_launch"] / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            break

    out = {
        "metric": "Mpixel*iterations/sec (Horn-Schunck: derivative pass + Jacobi u/v sweeps, frames resident in HBM)",
        "value": value, "unit": "Mpix*iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "warmup_steps_run": n_warm,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("a stream of %dx%d translating-texture pairs resident in HBM, a new pair every step (two seed pairs alternating), "
                                "lambda %g, %d Jacobi iterations, %s termination" % (W, H, args.lam, iters, head_name)) if pl else
                               ("%dx%d translating-texture pair(s), %d pair(s) per GPU per step, lambda %g, %d Jacobi iterations, %s termination"
                                % (W, H, pairs, args.lam, iters, head_name)),
                   "width": W, "height": H, "iters": iters, "pairs_per_gpu": pairs, "lambda": args.lam,
                   "kernel": KNAME[info["kernel"]],
                   "fuse_steps": info["fuse_steps"], "tile": [info["tile_w"], info["tile_h"]],
                   "threads": info["threads"], "rows_per_lane_or_groups": info["groups_per_thread"], "tiles_per_launch": info["tiles"],
                   "lds_bytes": info["lds_bytes"], "hipgraph": not args.no_graph, "termination": head_name,
                   "loop": loop,
                   "call": ("hsflow_pipeline_submit_device, %d slots on %d streams" % (args.stream_depth, min(args.stream_lanes, args.stream_depth))) if pl else
                           ("hsflow_solve" if args.sync_solves else "hsflow_solve_async"),
                   "iterations_done": info["iterations_done"], "eps_rerun": info["eps_rerun"],
                   "eps_check": "n/a (ITER)" if args.iter_only else (
                       "every pair's own check: witness words reduced and read when the pair's slot comes round again, a pair whose early stop fired "
                       "is re-solved from its slot's frames" if pl else "settled per solve (hsflow_solve)" if args.sync_solves else
                       "carried over identical repeats: every step solves the SAME resident pair, so a step takes over the early-stop "
                       "check the previous one owes and one check is settled after the timed region; a new pair per step is `fresh_frames`"),
                   "sharding": "independent pairs per rank, no collective"},
        "ms_per_step_blocks": block_ms, "ms_per_step_min": min(block_ms), "ms_per_step_median": statistics.median(block_ms),
        "ms_per_step_max": max(block_ms),
        "roofline": roof,
        "kernel_ms_per_step": {"deriv": der_ms / nprof, "jacobi": jac_ms / nprof, "launches": launches / nprof},
        "other_termination": {"termination": side_name, "ms_per_step": statistics.median(side_ms), "ms_per_step_blocks": side_ms,
                              "value": world * px * iters / (statistics.median(side_ms) * 1e-3) / 1e6,
                              "iterations_done": info2["iterations_done"], "eps_rerun": info2["eps_rerun"]},
    }
    if world > 1:
        out["rccl_ranks"] = dist.get_world_size()
        out["backend"] = backend

    # what a plain device-to-device copy reaches on this box (1 GiB read + 1 GiB written per copy,
    # torch's copy kernel, HIP events on torch's stream) -- context for the hbm figures
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
            roof["measured_copy_GBps"] = 2.0 * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except RuntimeError:
            pass
    # the reference's camera loop: a new pair every step, ITER|EPS, nothing carried over between steps
    if pl is not None:
        out["single_context"] = single
        if not args.iter_only:  # the headline IS that loop
            out["fresh_frames"] = {"is_the_headline": True, "ms_per_step": ms_per_step, "value": value, "unit": "Mpix*iter/s", "depth": args.stream_depth,
                                   "lanes": min(args.stream_lanes, args.stream_depth),
                                   "iterations_done": info["iterations_done"], "eps_rerun": info["eps_rerun"],
                                   "what": "a different resident pair every step, ITER|EPS (eps 1e-6), hsflow_pipeline_submit_device"}
        pl.close()
    elif not args.no_side and args.kernel == "auto" and pairs == 1 and not args.iter_only:
        try:
            out["fresh_frames"] = run_fresh_frames(args, hs, synth, torch, local_rank, W, H, iters, p_ieps, barrier, reduce_max, world)
        except hs.HsflowError as e:
            out["fresh_frames"] = {"error": str(e)}
    # the reference's own OpenCL discretisation (Kernels.cl, `-cl` route: 8-neighbour mean, alpha^2, IEEE division) on the
    # same frames, beside the headline: a side figure, never `value`
    pl_ok = loop == "stream"
    if rank == 0 and not args.no_side and args.kernel == "auto":
        try:
            pc = ctx.make_params(mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=iters, term_type=hs.TERM_ITER)
            for _ in range(5):
                ctx.solve_async(pc)
            ctx.synchronize()
            nc = max(5, min(args.steps, 50))
            t0 = time.perf_counter()
            for _ in range(nc):
                ctx.solve_async(pc)
            ctx.synchronize()
            cms = (time.perf_counter() - t0) / nc * 1e3
            ic = ctx.info()
            stream_ms = None
            if pl_ok:
                with hs.PairPipeline(W, H, depth=args.stream_depth, device=local_rank, lanes=min(args.stream_lanes, args.stream_depth)) as plc:
                    pcg = ctx.make_params(mode=hs.MODE_CLASSIC, alpha=15.0, max_iter=iters, term_type=hs.TERM_ITER, use_graph=True)

                    def goc(k):
                        for j in range(k):
                            plc.submit_device(dev_pairs[j & 1][0], dev_pairs[j & 1][1], params=pcg)
                        plc.drain()
                    goc(10)
                    t0 = time.perf_counter()
                    goc(nc)
                    stream_ms = (time.perf_counter() - t0) / nc * 1e3
            out["classic_mode"] = {"what": "Kernels.cl discretisation (v update restored), alpha 15, ITER, %d sweeps, same frames" % iters,
                                   "stream_ms_per_step": stream_ms, "stream_what": "a new resident pair every step through the pair pipeline (the headline's slots and streams)",
                                   "ms_per_step": cms, "value": px * iters / (cms * 1e-3) / 1e6, "unit": "Mpix*iter/s",
                                   "kernel": {hs.KERNEL_STRIP: "strip", hs.KERNEL_FUSED: "fused", hs.KERNEL_SIMPLE: "simple"}.get(ic["kernel"], str(ic["kernel"])),
                                   "fuse_steps": ic["fuse_steps"], "rows_per_lane_or_groups": ic["groups_per_thread"], "threads": ic["threads"],
                                   "launches": ic["jacobi_launches"]}
        except hs.HsflowError as e:
            out["classic_mode"] = {"error": str(e)}
    ctx.close()

    # --- multi-GPU configs of BASELINE.json beside the headline ---------------------------------------------
    want_c4 = args.c4 == "on" or (args.c4 == "auto" and world > 1)
    want_c5 = args.c5 == "on" or (args.c5 == "auto" and world > 1)
    # (a failure in one of these must not cost the headline: it is recorded under the key instead -- and neither must a
    # hang: the exchange of C5 is the one part of this file that has never run over RCCL on real peers, so a watchdog on
    # every rank ends the run after --side-timeout seconds; rank 0 prints the line with what it has first)
    watchdog = None
    if world > 1 and (want_c4 or want_c5) and args.side_timeout > 0:
        import threading

        def give_up():
            if rank == 0:
                late = dict(out)
                for k, want in (("c4_pipeline", want_c4), ("c5_slab", want_c5)):
                    if want and k not in late:
                        late[k] = {"error": "no result within --side-timeout %.0f s: abandoned, the headline stands" % args.side_timeout}
                sys.stdout.write(json.dumps(late) + "\n")
                sys.stdout.flush()
            os._exit(0)  # (a rank stuck inside a collective cannot be unwound; the process group dies with the process)

        watchdog = threading.Timer(args.side_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
    if want_c4:
        if args.debug_hang == "c4":
            time.sleep(1e6)
        try:
            out["c4_pipeline"] = run_c4(args, hs, synth, dist, world, rank, local_rank, barrier, reduce_max)
        except Exception as e:  # noqa: BLE001 -- reported in the line, the run goes on
            out["c4_pipeline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if want_c5:
        if args.debug_hang == "c5":
            time.sleep(1e6)
        try:
            out["c5_slab"] = run_c5(args, hs, synth, torch, dist, world, rank, local_rank, backend, barrier, reduce_max)
        except Exception as e:  # noqa: BLE001
            out["c5_slab"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0 and world == 1 and not args.no_side and args.kernel == "auto" and (W, H, iters) == (1920, 1080, 100):
        try:
            out["reference_default_workload"] = run_reference_default(args, hs, synth, torch, local_rank, with_cpu=not args.skip_cpu)
        except Exception as e:  # noqa: BLE001 -- a side figure must not cost the headline
            out["reference_default_workload"] = {"error": "%s: %s" % (type(e).__name__, e)}
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
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        import threading
        bye = threading.Timer(30.0, lambda: os._exit(0))  # a peer that gave up (above) never reaches this barrier: the line is out, leave
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
