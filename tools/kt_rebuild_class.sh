#!/bin/bash
# kernel trace of a short run of one of the smaller classes and the timeline of one list rebuild in it:
#   bash tools/kt_rebuild_class.sh TAG [WORKLOAD=dhfr23k]
TAG=${1:-ktrbc}; WL=${2:-dhfr23k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt_$WL" -o kt -- python3 bench.py --workload $WL --steps 300 --warmup 50 --no-cpu-baseline --no-extras --tail-steps 0 > "$OUT/bench_kt_$WL.json" 2> "$OUT/kt_$WL.err"
TIMELINE_BEFORE=8 python3 tools/rebuild_timeline.py "$OUT/kt_$WL" > "$OUT/timeline_$WL.txt" 2>&1
cat "$OUT/timeline_$WL.txt"
find "$OUT" -name "*.db" -delete
