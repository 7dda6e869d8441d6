#!/bin/bash
# kernel-trace summary of a short bench run.  Usage via gpurun: bash tools/kt.sh TAG [bench args]
TAG=${1:-kt}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
python3 tools/summarize_rocprof.py "$OUT" "$OUT" "$TAG"
head -32 "$OUT/${TAG}_rocprof_summary.txt"
find "$OUT" -name "*.db" -size +20M -delete
