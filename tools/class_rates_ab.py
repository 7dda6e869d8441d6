"""The driver line's `classes` block (bench.class_rate: no event brackets) under an environment knob the library reads per chunk, arms
interleaved in one process:  python tools/class_rates_ab.py MDX_ONEPASS 1 0 [rounds=2]"""
import os, sys
sys.path.insert(0, os.getcwd())
import bench
knob, vals = sys.argv[1], sys.argv[2:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for r in range(rounds):
    for v in vals:
        os.environ[knob] = v
        out = {w: bench.class_rate(w, 0.0005, 0) for w in ("dhfr23k", "complex50k", "dna100k")}
        print(f"{knob}={v}", "  ".join(f"{w} {o['steps_per_s']:.0f} ({o['rebuilds_so_far']} rebuilds)" for w, o in out.items()), flush=True)
