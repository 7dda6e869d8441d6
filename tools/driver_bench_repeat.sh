#!/bin/bash
# The driver's own bench command (20 timed steps behind 5 warm-up steps), repeated: how much the headline `value` moves with
# where the list rebuilds fall in the 20-step window.  bash tools/driver_bench_repeat.sh [N]
N=${1:-5}; O=$PWD/gpurun_out/drv; mkdir -p "$O"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/b1.json" 2>/dev/null
for i in $(seq 2 $N); do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$O/b$i.json" 2>/dev/null; done
python3 - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/b*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f.split("/")[-1], "steps/s %.1f" % d["steps_per_s"], "ms/step %.4f" % d["ms_per_step"], "rebuilds in window", d["config"].get("rebuilds_in_timed_region"),
          "| 1000-step tail %.1f" % (d.get("steps_per_s_1000") or 0), "| traffic", d["roofline"]["traffic"], "pair launch ms %.4f" % d["roofline"]["launch_ms"])
PY
