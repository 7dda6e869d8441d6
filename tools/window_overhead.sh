#!/bin/bash
# What a short timed window costs beyond its steps: bench.py at 20 / 40 / 80 / 160 timed steps, pair-kernel events on (2) and off (0)
O=$PWD/gpurun_out/win; mkdir -p "$O"
for pl in 2 0; do for n in 20 40 80 160; do for r in 1 2 3; do
  MDX_DEBUG_HOST=${DBG:-0} python3 bench.py --gpus 1 --steps $n --warmup 5 --settle-steps 0 --no-cpu-baseline --tail-steps 0 --profile-level $pl > "$O/b_${pl}_${n}_$r.json" 2> "$O/b_${pl}_${n}_$r.err"
done; done; done
python3 - "$O" <<'PY'
import json, glob, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json"), key=lambda s: [int(x) for x in re.findall(r"\d+", s.split("/")[-1])]):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    n = d["steps"]; rb = d["config"].get("rebuilds_in_timed_region")
    print(f.split("/")[-1], "ms/step %.4f" % d["ms_per_step"], "window ms %.3f" % (d["ms_per_step"] * n), "rebuilds", rb, "pair launch %.4f" % d["roofline"]["launch_ms"])
PY
