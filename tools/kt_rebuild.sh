#!/bin/bash
# kernel trace of a short bench run and of one rank of N, then the timeline of one list rebuild in each:
#   bash tools/kt_rebuild.sh TAG [WORLD=8] [SPLIT=0] [SKIP_1M=0]
TAG=${1:-ktrb}; W=${2:-8}; SPLIT=${3:-0}; SKIP=${4:-0}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
if [ "$SKIP" != "1" ]; then
rocprofv3 --kernel-trace -d "$OUT/kt1M" -o kt -- python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --tail-steps 0 > "$OUT/bench_kt.json" 2> "$OUT/kt1M.err"
python3 tools/rebuild_timeline.py "$OUT/kt1M" > "$OUT/timeline_1M.txt" 2>&1
fi
export ONE_RANK_TRACE=1 ONE_RANK_SPLIT=$SPLIT
rocprofv3 --kernel-trace -d "$OUT/kt$W" -o kt -- python3 tools/one_rank_profile.py $W 96 > "$OUT/one_rank$W.log" 2> "$OUT/kt$W.err"
python3 tools/rebuild_timeline.py "$OUT/kt$W" > "$OUT/timeline_rank${W}_split$SPLIT.txt" 2>&1
cat "$OUT"/timeline_*.txt
find "$OUT" -name "*.db" -delete
