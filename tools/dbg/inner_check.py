"""Step-loop (inner list) forces against a fresh plain-list evaluation of the same positions, step by step: python tools/dbg/inner_check.py [n_side=70] [steps=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
s = systems.water_box(n, seed=5)
with MdState(s, MdConfig()) as md:
    last_rb = md.stats()["rebuild_count"]
    for k in range(steps):
        md.step(0.0005, None, 1)
        fi = md.forces().astype(np.float64)
        st = md.stats(); info = md.pair_launch_info()
        md.energy()
        fp = md.forces().astype(np.float64)
        d = np.linalg.norm(fi - fp, axis=1)
        bad = int((d > 0.05).sum())
        print(f"step {k + 1}: rebuilds {st['rebuild_count']} prune passes {st['prune_passes']} inner-from-rebuild {info['inner_lists_from_rebuilds']} "
              f"max |dF| {d.max():.3e} atoms with |dF| > 0.05: {bad}", flush=True)
