#!/bin/bash
# kernel trace of one rank of N and the timeline of its steady-state steps: bash tools/kt_step.sh TAG WORLD SPLIT
TAG=${1:-ktstep}; W=${2:-8}; SPLIT=${3:-0}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
export ONE_RANK_TRACE=1 ONE_RANK_SPLIT=$SPLIT
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt -- python3 tools/one_rank_profile.py $W 96 > "$OUT/one_rank.log" 2> "$OUT/kt.err"
python3 tools/step_timeline.py "$OUT/kt" 12 all ${4:-0} > "$OUT/step_timeline_rank${W}_split$SPLIT.txt" 2>&1
cat "$OUT/step_timeline_rank${W}_split$SPLIT.txt"; tail -2 "$OUT/one_rank.log"
find "$OUT" -name "*.db" -delete
