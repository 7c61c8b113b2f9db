#!/bin/bash
# SQ stall / issue counters of bench.py's kernels (two passes of <= 8 SQ counters), on the GPU box.
# Outputs: gpurun_out/prof_sq_<tag>/{a,b}/...counter_collection.csv ; summarise with tools/collect_sq.py
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --skip-cpu --no-graph $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$OUT/a" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_a.json" 2> "$OUT/bench_a.err" || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT/b" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_b.json" 2> "$OUT/bench_b.err" || exit 2
find "$OUT" -name "*counter_collection.csv" | head
