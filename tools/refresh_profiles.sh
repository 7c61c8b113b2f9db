#!/bin/bash
# Runs on the GPU box (via gpurun): everything the numbers in DESIGN.md section 6 and profiles/ are quoted from.
# Outputs under gpurun_out/refresh/ ; copy into profiles/ with the names profiles/README.md lists.
# usage: tools/refresh_profiles.sh [tag]          (default tag r03)
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd "$ROOT"
run() { echo "== $*" >&2; timeout -k 10 300 "$@"; }
run python bench.py > "$OUT/bench_1080p.json" 2> "$OUT/bench_1080p.err" || exit 1
run python bench.py --steps 20 --warmup 5 --skip-cpu > "$OUT/bench_1080p_driver_flags.json" 2>> "$OUT/err.txt" || exit 2
run python bench.py --loop repeat --skip-cpu > "$OUT/bench_1080p_repeat_loop.json" 2>> "$OUT/err.txt" || exit 2
run python bench.py --sync-solves --skip-cpu > "$OUT/bench_1080p_sync_solves.json" 2>> "$OUT/err.txt" || exit 2
run python bench.py --width 3840 --height 2160 --iters 200 --steps 50 --warmup 5 --skip-cpu > "$OUT/bench_4k.json" 2>> "$OUT/err.txt" || exit 3
run python bench.py --width 3840 --height 2160 --iters 200 --steps 50 --warmup 5 --skip-cpu --stream-lanes 3 > "$OUT/bench_4k_3lanes.json" 2>> "$OUT/err.txt" || exit 3
{ run python tools/fresh_frames.py; run python tools/fresh_frames.py --width 3840 --height 2160 --iters 200 --steps 60; } > "$OUT/fresh_frames.txt" 2>> "$OUT/err.txt" || exit 3
{ HSFLOW_DEBUG_STAMPS=/tmp/persist_stamps.txt run python tools/persist_check.py --reps 200 --fuse-steps 20 16; run python tools/persist_check.py --width 424 --height 240 --reps 200 --fuse-steps 20; } > "$OUT/persist.txt" 2>> "$OUT/err.txt" || exit 3
run python tools/classic_stream.py > "$OUT/classic_stream.txt" 2>> "$OUT/err.txt" || exit 3
{
  echo "# 16 pairs per step (C4 shard of one GPU)";      run python bench.py --pairs 16 --steps 30 --warmup 5 --skip-cpu --loop repeat 2>> "$OUT/err.txt" || exit 4
  echo "# 16384 x 2048 strip (C5 slab of one GPU), 100 it"; run python bench.py --width 16384 --height 2048 --iters 100 --steps 10 --warmup 2 --skip-cpu 2>> "$OUT/err.txt" || exit 5
  echo "# simple kernel, 1080p";                           run python bench.py --kernel simple --steps 30 --warmup 5 --skip-cpu 2>> "$OUT/err.txt" || exit 6
  echo "# simple kernel, 4K/200";                          run python bench.py --kernel simple --width 3840 --height 2160 --iters 200 --steps 10 --warmup 2 --skip-cpu 2>> "$OUT/err.txt" || exit 7
  echo "# slab driver, one rank, 16384 x 2048, 100 it, halo 16 (tools/bench_slab.py)"
  run python tools/bench_slab.py --width 16384 --height 2048 --iters 100 --halo 16 --steps 3 2>> "$OUT/err.txt" || exit 8
} > "$OUT/bench_misc.txt"
{
  echo "# classic mode (Kernels.cl discretisation): the planner's choice (register strip), then the LDS-tile kernel, 1080p/100; then 4K/200"
  run python tools/bench_classic.py --steps 100 2>> "$OUT/err.txt" || exit 9
  run python tools/bench_classic.py --steps 100 --kernel fused 2>> "$OUT/err.txt" || exit 9
  run python tools/bench_classic.py --width 3840 --height 2160 --iters 200 --steps 20 2>> "$OUT/err.txt" || exit 9
  run python tools/bench_classic.py --width 3840 --height 2160 --iters 200 --steps 20 --kernel fused 2>> "$OUT/err.txt" || exit 9
} > "$OUT/bench_classic.txt"
run python tools/bench_e2e.py > "$OUT/e2e_pipeline.txt" 2>> "$OUT/err.txt" || exit 10
{ run python tools/crossover.py 100; run python tools/crossover.py 10; } > "$OUT/crossover.txt" 2>> "$OUT/err.txt" || exit 11
run python tools/time_cases.py > "$OUT/time_cases.txt" 2>> "$OUT/err.txt" || exit 12
run python tools/stamps.py --configs "20:5:1024;14:5:1024;12:4:1024" > "$OUT/phase_stamps.txt" 2>> "$OUT/err.txt" || exit 15
run env HSFLOW_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 2 --steps 20 --warmup 5 --c4-pairs 32 --c5-size 4096 --c5-iters 100 > "$OUT/bench_2rank_gloo_rehearsal.json" 2>> "$OUT/err.txt" || exit 16
run bash tools/pmc_by_kernel.sh "$TAG" 1920 1080 100 > "$OUT/pmc_1080p.log" 2>&1 || exit 13
run bash tools/pmc_by_kernel.sh "$TAG" 3840 2160 200 > "$OUT/pmc_4k.log" 2>&1 || exit 13
run bash tools/profile_gpu.sh "$TAG" > "$OUT/prof.log" 2>&1 || exit 13
run bash tools/profile_gpu.sh "${TAG}_4k" --width 3840 --height 2160 --iters 200 > "$OUT/prof_4k.log" 2>&1 || exit 14
run bash tools/profile_sq.sh "$TAG" --no-side --blocks 1 > "$OUT/prof_sq.log" 2>&1 || exit 17
echo done
