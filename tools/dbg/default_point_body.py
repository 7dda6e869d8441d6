"""Measures what tests/test_gpu_timed_body.py::test_default_operating_point_step_loop_forces_against_the_oracle bounds: the forces the
step loop leaves behind at the reference's default operating point (1,048,576 OPC sites, SPME, dt 2 fs) against the fp64 oracle
(real space) + the numpy SPME on the same mesh.  Run from the repo root on the GPU box: python tools/dbg/default_point_body.py [n_side=64]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from molchanica_amd import MdConfig, systems, _abi, md_state
from oracle import oracle as orc, pme_ref as P
from test_gpu_pme import excluded_pairs
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = systems.opc_water_box(n_side, seed=5)
beta = 0.3
cfg = MdConfig(coulomb_mode=2, ewald_alpha=beta, overrides=0, skin=2.0)
with md_state.MdState(s, cfg) as md:
    md.initialize_velocities(300.0, True, seed=1)
    md.step(0.002, None, steps)
    info, st = md.pair_launch_info(), md.stats()
    pos = md.positions(); f_step = md.forces().astype(np.float64)
    e = md.energy(); f_plain = md.forces().astype(np.float64)
print("launch info", info); print("rebuilds", st["rebuild_count"], "prune passes", st["prune_passes"], "fallbacks", st["rebuild_fallbacks"])
print({k: e[k] for k in ("lj", "coulomb", "coulomb_recip", "potential", "kinetic", "temperature")})
x = pos.astype(np.float64)
L = float(s.box_hi[0]); box = np.full(3, L); q = s.charge.astype(np.float64)
K = 1
while True:      # good_size: the 2-3-5-smooth mesh the library chooses at ~1 A
    K += 1
    k = K
    for p in (2, 3, 5):
        while k % p == 0: k //= p
    if k == 1 and K >= L: break
print("mesh", K)
t = time.time()
cfg_real = MdConfig(coulomb_mode=2, ewald_alpha=beta, overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED, skin=2.0)
fo, eo = orc.forces(s, cfg_real, pos=x, use_cells=True)
e_rec, f_rec = P.spme_recip(x, q, (0, 0, 0), box, beta, (K, K, K), 4)
e_x, f_x = P.excluded_pair_correction(x, q, excluded_pairs(s), box, beta)
e_rec += e_x + P.ewald_self_energy(q, beta) + P.ewald_background_energy(q, box, beta)
f_rec += f_x
vi, w = s.vsite_idx.astype(np.int64), s.vsite_w.astype(np.float64)
fm = f_rec[vi[:, 0]].copy()
f_rec[vi[:, 1]] += (1.0 - w[:, 0] - w[:, 1])[:, None] * fm; f_rec[vi[:, 2]] += w[:, 0:1] * fm; f_rec[vi[:, 3]] += w[:, 1:2] * fm; f_rec[vi[:, 0]] = 0.0
f_ref = fo + f_rec
slack = orc.cutoff_slack(s, cfg_real, pos=pos, rel=4e-5)
print("oracle seconds", time.time() - t)
rms = lambda a: math.sqrt((a ** 2).sum(1).mean())
for name, f in (("step-loop", f_step), ("plain-list", f_plain)):
    err = np.linalg.norm(f - f_ref, axis=1)
    tol = 1e-4 * np.maximum(np.linalg.norm(f_ref, axis=1), 1.0) + slack
    print(name, "rms err", rms(f - f_ref), "rel to rms f_rec", rms(f - f_ref) / rms(f_rec), "rel to rms f", rms(f - f_ref) / rms(f_ref), "max err", err.max(), "max err/tol(1e-4)", (err / tol).max(),
          "atoms over tol", int((err > tol).sum()), "M-site force", np.abs(f[3::4]).max())
print("step vs plain: rms", rms(f_step - f_plain), "max", np.abs(f_step - f_plain).max())
print("energies: lj", e["lj"], eo["lj"], "coulomb", e["coulomb"], eo["coulomb"], "recip", e["coulomb_recip"], e_rec)
