#!/bin/bash
# BASELINE config C3's tile sweep with the counter column: the strip kernel's shapes at 3840x2160 / 200
# sweeps, timed in one process (tools/sweep.py), then the five fastest re-run under rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE (separate passes) -> gpurun_out/sweep_counters_<tag>/ ; summarised by tools/collect_sweep_counters.py
set -o pipefail
TAG=${1:-r02_4k}; W=${2:-3840}; H=${3:-2160}; IT=${4:-200}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_counters_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 tools/sweep.py --width $W --height $H --iters $IT --no-fused --strip 4,5,6 --waves 12,16 --fuse 10,12,14,16,18,20,22,24 --rounds 3 --reps 5 --tag $TAG > "$OUT/sweep.txt" 2>&1 || exit 1
tail -20 "$OUT/sweep.txt"
python3 - "$ROOT/gpurun_out/sweep_$TAG.csv" > "$OUT/top5.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["name"].startswith("S_")][:5]
for r in rows:
    print(r["name"], r["T"], r["k"], r["threads"], r["median_ms"], r["tiles"], r["tile_w"], r["tile_h"])
PY
cat "$OUT/top5.txt"
cd /tmp && export TMPDIR=/tmp
while read -r name T R NT ms tiles tw th; do
  ARGS="--width $W --height $H --iters $IT --steps 3 --warmup 1 --blocks 1 --skip-cpu --no-side --iter-only --no-graph --kernel strip --fuse-steps $T --strip-rows $R --threads $NT"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/$name/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.fetch.json" 2> "$OUT/$name.fetch.err" || exit 2
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/$name/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.write.json" 2> "$OUT/$name.write.err" || exit 3
  echo "$name done"
done < "$OUT/top5.txt"
