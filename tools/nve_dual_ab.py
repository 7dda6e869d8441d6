"""NVE energy drift of the 1M-atom box with the dual pair list off and on (same preparation as tools/long_run.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
s = systems.BY_NAME["water1M"]()
for inner in (-1.0, 0.0):
    with MdState(s, MdConfig(coulomb_mode=1, inner_skin=inner)) as md:
        md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=5)
        md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.0005, None, 600); md.set_thermostat(0, 300.0, 0.02, 1)
        e0 = md.energy()
        out = []
        for k in range(6):
            md.step(0.0005, None, 500)
            e = md.energy()
            out.append(round((e["potential"] + e["kinetic"] - e0["potential"] - e0["kinetic"]) / s.n_atoms, 6))
        print("inner_skin", inner, "dE/atom per 500 steps:", out, "prunes", md.stats()["prune_passes"])
