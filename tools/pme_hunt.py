import sys; sys.path.insert(0, ".")
import numpy as np
from molchanica_amd import md_state as mdx, systems, MdConfig
s = systems.water1m()
cfgs = {"pme": MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0), "real": MdConfig(coulomb_mode=2, ewald_alpha=0.3), "cutoff": MdConfig()}
with mdx.MdState(s, cfgs["pme"]) as md:
    md.minimize_energy(100)
    x = md.positions(); f = md.forces(); fn = np.linalg.norm(f, axis=1); j = int(fn.argmax())
print("after minimise: Fmax %.1f atom %d, pos %s" % (fn[j], j, x[j]))
m = j // 3 * 3
print("molecule atoms", m, m + 1, m + 2, "OH", np.linalg.norm(x[m + 1] - x[m]), np.linalg.norm(x[m + 2] - x[m]), "HH", np.linalg.norm(x[m + 1] - x[m + 2]))
# nearest neighbours of the molecule's atoms
for a in (m, m + 1, m + 2):
    d = x - x[a]; L = float(s.box_hi[0]); d -= np.round(d / L) * L
    r = np.linalg.norm(d, axis=1); o = np.argsort(r)[1:6]
    print("  atom", a, "nearest:", [(int(k), round(float(r[k]), 3)) for k in o])
s.pos = x
for name, cfg in cfgs.items():
    with mdx.MdState(s, cfg) as md:
        f0 = md.forces()[j].astype(np.float64)
        num = np.zeros(3)
        for k in range(3):
            es = []
            for sgn in (+1, -1):
                xp = x.copy(); xp[j, k] += sgn * 0.02
                md.set_positions(xp); es.append(md.energy()["potential"])
            num[k] = -(es[0] - es[1]) / 0.04
        print(name, "force on atom", j, f0, " -dE/dx", num)
