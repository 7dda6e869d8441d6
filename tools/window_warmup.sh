#!/bin/bash
# Does the 20-step window of the driver's command pay for things a longer warm-up would have paid?  bench.py --steps 20 at several --warmup
O=$PWD/gpurun_out/winw; mkdir -p "$O"
for w in 5 30 100 300; do for r in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup $w --settle-steps 0 --no-cpu-baseline --tail-steps 0 > "$O/b_${w}_$r.json" 2> /dev/null
done; done
python3 - "$O" <<'PY'
import json, glob, sys, re
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json"), key=lambda s: [int(x) for x in re.findall(r"\d+", s.split("/")[-1])]):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f.split("/")[-1], "warmup", d["warmup"], "ms/step %.4f" % d["ms_per_step"], "steps/s %.1f" % d["steps_per_s"], "rebuilds", d["config"].get("rebuilds_in_timed_region"))
PY
