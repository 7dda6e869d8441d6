"""Diagnose the worst per-atom force difference between the HIP path and the oracle on water1M."""
import sys, numpy as np
sys.path.insert(0, ".")
from molchanica_amd import md_state as mdx, systems
from molchanica_amd._abi import MdConfig
from oracle import oracle as orc
s = systems.water1m(); cfg = MdConfig()
print("cfg", cfg)
with mdx.MdState(s, cfg) as md:
    f = md.forces().astype(np.float64); pos = md.positions()
fo, eo = orc.forces(s, cfg, pos=pos.astype(np.float64), use_cells=True)
for rel in (1e-5, 4e-5, 1e-3):
    sl = orc.cutoff_slack(s, cfg, pos=pos, rel=rel)
    err = np.linalg.norm(f - fo, axis=1)
    tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + sl
    r = err / tol
    w = np.argsort(r)[-5:][::-1]
    print("rel", rel, "nslack", np.count_nonzero(sl))
    for i in w:
        print("  atom", i, "ratio %.3f err %.4e slack %.4e |fo| %.3f q %.3f pos" % (r[i], err[i], sl[i], np.linalg.norm(fo[i]), s.charge[i]), pos[i], "dF", f[i] - fo[i])
i = int(np.argmax(err))
d = pos - pos[i]; L = s.box_hi[0] - s.box_lo[0]
d -= np.rint(d / L) * L
r2 = (d.astype(np.float64) ** 2).sum(1)
near = np.nonzero(np.abs(r2 - 100.0) < 1e-2)[0]
for j in near:
    print("   j", j, "r2 %.8f q %.3f" % (r2[j], s.charge[j]), "err_j %.3e" % err[j])
