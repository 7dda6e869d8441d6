import os, sys, math
import numpy as np
sys.path.insert(0, os.getcwd())
from molchanica_amd import MdConfig, systems, md_state
s = systems.small_solvated()
cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
L = np.array(s.box_hi) - np.array(s.box_lo)
def run(op):
    os.environ["MDX_ONEPASS"] = op
    out = []
    with md_state.MdState(s, cfg) as md:
        md.set_snapshot_cadence(10, with_velocities=True)
        md.step(0.0005, None, 10)
        out.append((md.positions().astype(np.float64), md.velocities().astype(np.float64), md.forces().astype(np.float64)))
        for n in [int(x) for x in os.environ.get('CALLS','1,1,1,3').split(',')]:
            md.step(0.0005, None, n)
            out.append((md.positions().astype(np.float64), md.velocities().astype(np.float64), md.forces().astype(np.float64)))
        st = md.stats()
        print(op, "rebuilds", st["rebuild_count"], "prunes", st["prune_passes"], md.pair_launch_info()["one_launch_steps"], md.pair_launch_info()["kicks_beyond_grant"])
    return out
a, b = run("1"), run("0")
for k, ((xa, va, fa), (xb, vb, fb)) in enumerate(zip(a, b)):
    d = xa - xb; d -= np.round(d / L) * L
    print(k, "pos rms", f"{math.sqrt((d**2).sum(1).mean()):.2e}", "vel rms", f"{math.sqrt(((va-vb)**2).sum(1).mean()):.2e}", "force max", f"{np.abs(fa-fb).max():.2e}",
          "worst atom", int(np.abs(va-vb).sum(1).argmax()), "n bad", int((np.abs(va-vb).max(1) > 0.1).sum()))
