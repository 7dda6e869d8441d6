"""Development smoke: GPU vs oracle diagnostics (verbose)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState, device_count
from oracle import oracle

print("devices", device_count())

def cmp_forces(name, s, cfg):
    t = time.time()
    md = MdState(s, cfg)
    print(name, "create %.3fs" % (time.time() - t), md.stats())
    pos = md.positions()
    f = md.forces().astype(np.float64)
    e = md.energy()
    fo, eo = oracle.forces(s, cfg, pos=pos.astype(np.float64), use_cells=s.n_atoms > 3000)
    slack = oracle.cutoff_slack(s, cfg, pos=pos) if s.periodic else np.zeros(s.n_atoms)
    df = np.linalg.norm(f - fo, axis=1)
    tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + slack
    print(name, "max|dF| %.3e  rms dF/rms F %.3e  worst ratio %.3f  nslack %d" % (
        df.max(), np.sqrt((df**2).mean()) / np.sqrt((fo**2).sum(1).mean()), (df / tol).max(), (slack > 0).sum()))
    for k in ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14"):
        print("   %-10s gpu %.6f  orc %.6f  d %.2e" % (k, e[k], eo[k], e[k] - eo[k]))
    return md

s = systems.lig50()
md = cmp_forces("lig50", s, MdConfig(lj_cutoff=0, coulomb_cutoff=0))
s2 = systems.small_solvated()
cfg2 = MdConfig(lj_cutoff=9, coulomb_cutoff=9, skin=1.5)
md2 = cmp_forces("solv120", s2, cfg2)
# neighbour list
off, idx = md2.neighbor_list()
pos = md2.positions()
ooff, oidx = oracle.neighbor_list(s2, 10.5, pos=pos)
print("nlist equal:", np.array_equal(off, ooff) and np.array_equal(idx, oidx), off[-1], ooff[-1])
# trajectory
x0 = md2.positions().astype(np.float64); v0 = md2.velocities().astype(np.float64)
md2.step(0.0005, None, 100)
xg = md2.positions().astype(np.float64)
xo, vo, eo = oracle.step(s2, cfg2, 0.0005, 100, pos=x0, vel=v0, use_cells=False)
L = np.array(s2.box_hi) - np.array(s2.box_lo)
d = xg - xo; d -= np.round(d / L) * L
print("traj rms dev %.3e max %.3e  rebuilds %d" % (np.sqrt((d**2).sum(1).mean()), np.abs(d).max(), md2.stats()["rebuild_count"]))
print("energy after", md2.energy()["potential"], eo["potential"])

# water box timing
for n in (16, 40):
    s3 = systems.water_box(n)
    t = time.time(); md3 = MdState(s3, MdConfig()); print("water", s3.n_atoms, "create %.2fs" % (time.time() - t), md3.stats())
    md3.step(0.0005, None, 20)
    md3.profile(True)
    t = time.time(); md3.step(0.0005, None, 100); dt = time.time() - t
    st = md3.stats()
    print("water %d: %.1f steps/s  nb %.3f ms  bonded %.3f ms  integ %.3f ms rebuilds %d rebuild_ms %.2f" % (
        s3.n_atoms, 100 / dt, st["nb_ms_sum"] / max(st["nb_launches"], 1), st["bonded_ms_sum"] / max(st["bonded_launches"], 1),
        st["integ_ms_sum"] / max(st["integ_launches"], 1), st["rebuild_count"], st["rebuild_ms_sum"]))
    print(md3.energy())
