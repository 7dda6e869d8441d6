#!/bin/bash
# A/B of one environment knob on the bench line, arms interleaved on ONE box (box-to-box variance is ~3 %):
#   bash tools/env_ab.sh MDX_ROLE_HEADS "1 0" [rounds] [bench args]
# prints: KNOB=v steps/s ms/step fused-pass-ms pair-ms rebuild-ms-per-step rebuilds
K=$1; VALS=${2:-"1 0"}; R=${3:-2}; shift 3 2>/dev/null
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do for v in $VALS; do
  env $K=$v python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras "$@" > gpurun_out/ab/bench_${K}_$v.json 2>/dev/null
  python3 - "$K=$v" gpurun_out/ab/bench_${K}_$v.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
km = j.get("kernel_ms") or {}
print(sys.argv[1], round(j["steps_per_s"], 1), round(j["ms_per_step"], 5), km.get("bonded_integrate_fused"), km.get("nonbonded"),
      round(j["rebuild_ms_per_step_amortised"], 5), j["config"].get("rebuilds_in_timed_region"))
PY
done; done
