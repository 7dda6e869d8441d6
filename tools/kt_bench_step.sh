#!/bin/bash
# kernel trace of bench.py at a workload and the timeline of its steady-state steps: bash tools/kt_bench_step.sh TAG WORKLOAD [after]
TAG=${1:-ktb}; WL=${2:-dhfr23k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt -- python3 bench.py --workload $WL --steps 600 --warmup 50 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/kt.err"
STEP_TL_END_FRAC=0.6 python3 tools/step_timeline.py "$OUT/kt" 12 all ${3:-40} > "$OUT/step_timeline_$WL.txt" 2>&1
python3 tools/rebuild_timeline.py "$OUT/kt" -12 > "$OUT/rebuild_timeline_$WL.txt" 2>&1
cat "$OUT/step_timeline_$WL.txt"; head -40 "$OUT/rebuild_timeline_$WL.txt"
python3 - "$OUT/bench.json" <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: b.get(k) for k in ("steps_per_s", "ms_per_step", "rebuild_ms_per_step_amortised", "rebuilds_in_timed_region")})
PY
find "$OUT" -name "*.db" -delete
