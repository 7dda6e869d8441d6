import sys; sys.path.insert(0, ".")
import numpy as np
from molchanica_amd import md_state as mdx, systems, MdConfig
s = systems.water1m()
which = sys.argv[1]
v = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0, nb_variant=v) if which == "pme" else MdConfig(nb_variant=v)
with mdx.MdState(s, cfg) as md:
    md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=105)
    md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.0005, None, 600); md.set_thermostat(0, 300.0, 0.02, 1)
    def rb(): return md.stats()["rebuild_count"]
    out = []
    for k in range(4):
        r = rb(); md.step(0.0005, None, 100); out.append(rb() - r)
    e = md.energy()
    print(which, "rebuilds per 100 steps:", out, "T %.1f" % e["temperature"], "maxF %.1f" % e["max_force"])
