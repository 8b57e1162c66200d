#!/bin/bash
# Collects the rocprofv3 evidence for one round (run on the GPU box through gpurun):
#   1. --kernel-trace --stats  of the default bench command  -> per-kernel average durations
#   2. --pmc FETCH_SIZE        (own pass, kernel-trace only) -> HBM read bytes per launch
#   3. --pmc WRITE_SIZE        (own pass)                    -> HBM write bytes per launch
# Usage: profiles/run_profiles.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
cd "$ROOT"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-back-to-back $*"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err" || { echo "stats pass: non-zero exit"; tail -30 "$OUT/stats.err"; exit 1; }
ls "$OUT"/stats/*/*_kernel_stats.csv > /dev/null 2>&1 || { echo "stats pass produced no output"; tail -5 "$OUT/stats.err"; exit 1; }
timeout -k 10 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || { echo "fetch pass: non-zero exit"; tail -30 "$OUT/fetch.err"; exit 1; }
ls "$OUT"/pmc_fetch/*/*_counter_collection.csv > /dev/null 2>&1 || { echo "fetch pass produced no output"; tail -5 "$OUT/fetch.err"; exit 1; }
timeout -k 10 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err" || { echo "write pass: non-zero exit"; tail -30 "$OUT/write.err"; exit 1; }
ls "$OUT"/pmc_write/*/*_counter_collection.csv > /dev/null 2>&1 || { echo "write pass produced no output"; tail -5 "$OUT/write.err"; exit 1; }
python3 profiles/summarize.py "$OUT" "$TAG" > "$OUT/summary_$TAG.md"
cat "$OUT/summary_$TAG.md"
# keep only the small artefacts (the merged gpurun_out is capped at 64 MiB)
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*_counter_collection.csv" -size +8M -delete
