"""Decomposed (2 / 4 / 8 virtual ranks on one GPU, in-process fabric) against single-GPU runs FROM EQUILIBRATED, WRAPPED
STATES: energies at the start, trajectory deviation after 40 steps, energy conservation over 400 steps.  The lattices of whole
molecules the generators produce hide what atom-wise wrapping does to clusters owned across a periodic face (round 2: ghost
copies of straddling rigid waters a box length off).  Usage (through gpurun): python tools/decomp_soak.py"""
import sys, os, dataclasses, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
from tests.test_gpu_comm import run_ranks, rms_dev

def case(name, s, cfg, dt, n_eq=2000, n_cmp=40, n_nve=400):
    with MdState(s, cfg) as md:
        md.minimize_energy(200); md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.05, 1); md.step(min(dt, 0.001), None, n_eq)
        md.set_thermostat(0, 300.0, 0.05, 1)
        pos, vel = md.positions(), md.velocities()
    s2 = dataclasses.replace(s, pos=pos, vel=vel)
    with MdState(s2, cfg) as md:
        e0 = md.energy(); md.step(dt, None, n_cmp); p_ref = md.positions().astype(np.float64); e1 = md.energy()
        md.step(dt, None, n_nve - n_cmp); e2 = md.energy()
    L = np.array(s.box_hi, dtype=np.float64)
    t = lambda e: e["potential"] + e["kinetic"]
    print(f"{name}: single GPU dE over {n_nve} steps {t(e2) - t(e0):8.2f} (E_kin {e0['kinetic']:.0f})", flush=True)
    for world in (2, 4, 8):
        res = run_ranks(s2, cfg, world, n_cmp, dt=dt)
        r0 = res[0]
        res2 = run_ranks(s2, cfg, world, n_nve, dt=dt)
        q = res2[0]
        print(f"   world {world}: e0 diff pot {r0['e0']['potential'] - e0['potential']:8.3f}  rms dev after {n_cmp} steps {rms_dev(r0['pos'], p_ref, L):.2e} A  dE over {n_nve} steps {t(q['e1']) - t(q['e0']):8.2f}  T {q['e1']['temperature']:.1f}  repartitions {q['stats']['repartitions']} local rebuilds {q['stats']['local_rebuilds']}", flush=True)

rf = dict(coulomb_mode=1, lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5)
if len(sys.argv) > 1 and sys.argv[1] == "water1M":
    # the headline box at full size on 2 / 4 / 8 virtual ranks (fp32 coordinates in a 217 A box resolve 1.3e-5 A: two runs that
    # round in different frames drift apart by ~1e-4 A in 60 steps whatever else they do)
    case("water1M flexible TIP3P, shifted cutoff (bench configuration)", systems.BY_NAME["water1M"](), MdConfig(), 0.0005, n_eq=600, n_cmp=60, n_nve=300)
    sys.exit(0)
case("rigid OPC 16k sites RF dt 2 fs", systems.opc_water_box(16, seed=3), MdConfig(**rf), 0.002)
case("flexible TIP3P 17k RF", systems.water_box(18, seed=5), MdConfig(**rf), 0.0005)
case("solvated chain 400 RF", systems.small_solvated(n_chain=400, box=44.0), MdConfig(**rf), 0.0005)
case("rigid OPC 16k sites SPME dt 2 fs", systems.opc_water_box(16, seed=3), MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0, lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5), 0.002)
