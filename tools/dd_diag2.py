import sys, os, math, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import MdConfig, systems
from molchanica_amd.md_state import MdState, Fabric
from tools.dd_diag import run_ranks

def cmp(s, cfg, dt, nsteps, world=2, per=4):
    L = np.array(s.box_hi, dtype=np.float64)
    with MdState(s, cfg) as md:
        md.step(dt, None, nsteps)
        p_ref = md.positions().astype(np.float64); rb = md.stats()["rebuild_count"]
    res = run_ranks(s, cfg, world, nsteps, dt)
    d = res[0]["pos"].astype(np.float64) - p_ref
    d -= np.round(d / L) * L
    n = np.linalg.norm(d, axis=1)
    top = np.argsort(-n)[:6]
    print(f"  steps {nsteps}: rms {math.sqrt((n**2).mean()):.2e} max {n.max():.2e} rebuilds ref {rb} dd {res[0]['stats']['rebuild_count']} reparts {res[0]['stats']['repartitions']} "
          f"top: " + " ".join(f"{i}(k{i%per},x={p_ref[i,0]:.1f},{n[i]:.1e})" for i in top))

cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1, chunk_steps=8)
print("rigid TIP3P (no virtual sites), dt 2 fs")
s = systems.water_box(16, seed=3, rigid=True)
for n in (1, 2, 4, 8, 20): cmp(s, cfg, 0.002, n, per=3)
print("OPC, dt 2 fs")
s = systems.opc_water_box(16, seed=3)
for n in (1, 2, 3, 4, 6, 8): cmp(s, cfg, 0.002, n)
print("OPC, dt 2 fs, chunk_steps 1")
cfg1 = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1, chunk_steps=1)
for n in (4, 8): cmp(s, cfg1, 0.002, n)
