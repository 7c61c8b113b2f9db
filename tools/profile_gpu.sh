#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + HBM traffic counters of bench.py.
# Outputs land in gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by tools/collect_profiles.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --loop repeat: one solve after the other on ONE stream, so that a kernel's duration in the trace is its duration alone on
# the chip -- what bench.py's roofline prices (HIP events around eager launches); in the default stream loop two slots'
# launches overlap and every kernel's wall time stretches.
ARGS="--steps 20 --warmup 3 --blocks 2 --skip-cpu --loop repeat --no-side $*"   # (--no-side: no side figures on other frame sizes / streams in the trace)
# 1) per-kernel durations (the command bench.py's roofline line is checked against)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err" || exit 1
# 2) HBM traffic, one counter group per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS --no-graph > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS --no-graph > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || exit 3
find "$OUT" -name "*.csv" | head -20
