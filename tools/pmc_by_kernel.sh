#!/bin/bash
# HBM bytes per launch BY KERNEL VARIANT (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md's HBM section prescribes).  Runs on the GPU box; summary by tools/pmc_by_kernel.py.
# usage: tools/pmc_by_kernel.sh <tag> W H ITERS
set -o pipefail
TAG=${1:-r03}; W=${2:-1920}; H=${3:-1080}; IT=${4:-100}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_${TAG}_${W}x${H}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/pmc_workload.py" $W $H $IT > "$OUT/fetch.log" 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/pmc_workload.py" $W $H $IT > "$OUT/write.log" 2>&1 || exit 3
python3 "$ROOT/tools/pmc_by_kernel.py" "$OUT" $W $H > "$ROOT/gpurun_out/${TAG}_traffic_by_kernel_${W}x${H}.json" || exit 4
cat "$ROOT/gpurun_out/${TAG}_traffic_by_kernel_${W}x${H}.json"
