"""Which pairs are behind the atoms whose step-loop force differs from the oracle's at water1M?  python tools/dbg/c5_outliers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
from oracle import oracle as orc
orc.lib()
s = systems.water1m(); cfg = MdConfig()
L = np.asarray(s.box_hi, np.float64) - np.asarray(s.box_lo, np.float64)
with MdState(s, cfg) as md:
    md.step(0.0005, None, 20); md.energy(); md.step(0.0005, None, 12)
    pos = md.positions(); fs = md.forces().astype(np.float64)
    md.energy(); fp = md.forces().astype(np.float64)
fo, eo = orc.forces(s, cfg, pos=pos.astype(np.float64), use_cells=True)
for rel in (1e-5, 4e-5, 1e-4, 4e-4):
    slack = orc.cutoff_slack(s, cfg, pos=pos, rel=rel)
    tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + slack
    for name, f in (("step-loop", fs), ("plain", fp)):
        r = np.linalg.norm(f - fo, axis=1) / tol
        print(f"slack band {rel:.0e}: {name}: worst {r.max():.2f}, atoms above 1: {int((r > 1).sum())}, atoms with slack {int((slack > 0).sum())}")
slack = orc.cutoff_slack(s, cfg, pos=pos, rel=4e-5)
tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + slack
err = np.linalg.norm(fs - fo, axis=1)
x = pos.astype(np.float64)
for w in np.argsort(-(err / tol))[:4]:
    d = x[w] - x; d -= np.round(d / L) * L
    r = np.sqrt((d ** 2).sum(1))
    near = np.nonzero(np.abs(r - 10.0) < 2e-3)[0]
    dF = fs[w] - fo[w]
    print(f"atom {w} (type {'O' if w % 3 == 0 else 'H'}): |dF| {err[w]:.4f} tol {tol[w]:.4f} slack {slack[w]:.4f} dF {dF}; plain-list |dF| {np.linalg.norm(fp[w] - fo[w]):.4f}")
    for j in near:
        qq = 332.0637 * float(s.charge[w]) * float(s.charge[j])
        fpair = qq / r[j] ** 2 * d[j] / r[j]
        # fp32 arithmetic of the same distance, wrapped positions
        df = (pos[w] - pos[j]).astype(np.float32); df = df - np.rint(df / L.astype(np.float32)) * L.astype(np.float32)
        r2f = np.float32(df[0] * df[0]) + np.float32(df[1] * df[1]) + np.float32(df[2] * df[2])
        print(f"    partner {j} ({'O' if j % 3 == 0 else 'H'}): r {r[j]:.7f} (r^2/rc^2 - 1 = {r[j] ** 2 / 100 - 1:+.2e}; fp32 wrapped {float(r2f) / 100 - 1:+.2e}) pair force {fpair} |{np.linalg.norm(fpair):.4f}|  pos_w {pos[w]} pos_j {pos[j]}")
