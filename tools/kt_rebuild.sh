#!/bin/bash
# kernel trace of a short bench run and of one rank of 8, then the timeline of one list rebuild in each: bash tools/kt_rebuild.sh TAG
TAG=${1:-ktrb}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt1M" -o kt -- python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --tail-steps 0 > "$OUT/bench_kt.json" 2> "$OUT/kt1M.err"
python3 tools/rebuild_timeline.py "$OUT/kt1M" > "$OUT/timeline_1M.txt" 2>&1
export ONE_RANK_TRACE=1 ONE_RANK_SPLIT=0
rocprofv3 --kernel-trace -d "$OUT/kt8" -o kt -- python3 tools/one_rank_profile.py 8 96 > "$OUT/one_rank8.log" 2> "$OUT/kt8.err"
python3 tools/rebuild_timeline.py "$OUT/kt8" > "$OUT/timeline_rank8.txt" 2>&1
cat "$OUT/timeline_1M.txt" "$OUT/timeline_rank8.txt"
find "$OUT" -name "*.db" -delete
