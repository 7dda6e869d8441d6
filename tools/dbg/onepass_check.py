"""One launch per step against the fused-pass arrangement on a small box: python tools/dbg/onepass_check.py  (run with MDX_WPT=1 MDX_WPT8_BELOW=32;
MDX_ONEPASS=0 for the other arm)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from molchanica_amd import MdConfig, systems, md_state
s = systems.water_box(int(os.environ.get("N_SIDE", "16")), seed=41)
with md_state.MdState(s, MdConfig()) as md:
    for burst in (7, 20, 33, 100, 200):
        md.step(0.0005, None, burst)
        st, info = md.stats(), md.pair_launch_info()
        e = md.energy()
        print(burst, "rebuilds", st["rebuild_count"], "prunes", st["prune_passes"], "one-launch", info["one_launch_steps"], "beyond grant", info["kicks_beyond_grant"],
              "dual", info["step"]["dual"], "E", round(e["potential"] + e["kinetic"], 3), "T", round(e["temperature"], 1))
