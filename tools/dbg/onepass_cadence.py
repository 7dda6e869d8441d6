"""One launch per step against the separate passes, with and without an energy cadence (chunks that end with an energy evaluation)."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.getcwd())
from molchanica_amd import MdConfig, systems, md_state
s = systems.small_solvated()
cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
L = np.array(s.box_hi) - np.array(s.box_lo)
def run(op, cadence, calls):
    os.environ["MDX_ONEPASS"] = op
    with md_state.MdState(s, cfg) as md:
        if cadence: md.set_snapshot_cadence(cadence, with_velocities=True)
        out = []
        for n in calls:
            md.step(0.0005, None, n)
            out.append((md.positions().astype(np.float64), md.velocities().astype(np.float64)))
        info = md.pair_launch_info()
    return out, info
for cadence, calls in ((0, (10, 10, 10)), (10, (30,)), (10, (10, 10, 10)), (0, (30,)), (0, (1,) * 12)):
    a, ia = run("1", cadence, calls); b, ib = run("0", cadence, calls)
    for k, ((xa, va), (xb, vb)) in enumerate(zip(a, b)):
        d = xa - xb; d -= np.round(d / L) * L
        print("cadence", cadence, "calls", calls[:3], "after call", k, "pos rms", f"{math.sqrt((d**2).sum(1).mean()):.2e}", "vel rms", f"{math.sqrt(((va-vb)**2).sum(1).mean()):.2e}",
              "one-launch steps", ia["one_launch_steps"], ib["one_launch_steps"])
