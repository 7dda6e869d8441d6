#!/bin/bash
# timeline of steady-state steps at the reference's default operating point: bash tools/kt_default_point_steps.sh TAG [N_SIDE=18]
TAG=${1:-kt_dpsteps}; NS=${2:-18}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp DP_ONLY=spme
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt -- python3 tools/default_point_time.py $NS > "$OUT/run.log" 2> "$OUT/kt.err"
python3 tools/step_timeline.py "$OUT/kt" 6 all > "$OUT/steps.txt" 2>&1
cat "$OUT/steps.txt"; tail -1 "$OUT/run.log"
find "$OUT" -name "*.db" -delete
