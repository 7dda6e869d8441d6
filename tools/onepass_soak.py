"""Soak of one launch per step: the three small classes, 200,000 NVE steps each (reaction field: the shifted cutoff drifts by construction), and a
hot run (600 K); reports energy drift, list rebuilds, pruning passes and the steps taken back (launches that contradicted their gating words).
python tools/onepass_soak.py  ->  profiles/rNN_onepass_soak.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from molchanica_amd import MdConfig, systems, md_state
def run(name, temp, n, block=20000):
    s = systems.BY_NAME[name]()
    cfg = MdConfig(coulomb_mode=1)
    with md_state.MdState(s, cfg) as eq:
        eq.minimize_energy(100); eq.initialize_velocities(temp, True, seed=105)
        eq.set_thermostat(1, temp, 0.02, 1); eq.step(0.0005, None, 2000); eq.set_thermostat(0, temp, 0.02, 1)
        s.pos, s.vel = np.ascontiguousarray(eq.positions(), np.float32), np.ascontiguousarray(eq.velocities(), np.float32)
    with md_state.MdState(s, cfg) as md:
        e0 = md.energy(); t0 = time.time(); es = []
        for k in range(n // block):
            md.step(0.0005, None, block)
            e = md.energy(); es.append(e["potential"] + e["kinetic"])
        el = time.time() - t0
        st, info = md.stats(), md.pair_launch_info()
    tot0 = e0["potential"] + e0["kinetic"]
    drift = (es[-1] - tot0) / (n / 1000.0)
    print(f"{name:11s} {temp:5.0f} K  {n} steps in {el:5.1f} s ({n / el:7.0f} steps/s incl. {n // block} energy reads)  E_tot {tot0:12.1f}  drift {drift:8.3f} kcal/mol per 1000 steps "
          f"= {100 * drift / e0['kinetic']:7.4f} % of E_kin  T {e0['temperature']:.1f} -> {e['temperature']:.1f} K  rebuilds {st['rebuild_count']}  prunes {st['prune_passes']}  "
          f"one-launch steps {info['one_launch_steps']}  taken back {info['kicks_beyond_grant']}", flush=True)
for name in ("dhfr23k", "complex50k", "dna100k"):
    run(name, 300.0, 200000)
run("dhfr23k", 600.0, 100000)
