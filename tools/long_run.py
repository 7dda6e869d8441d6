"""Longer NVE / NVT / NPT runs of the current engine: energy drift, temperature, blow-up watch.
Usage: python tools/long_run.py water1M|dhfr23k|rigid  [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
name = sys.argv[1] if len(sys.argv) > 1 else "dhfr23k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
if name == "rigid":
    s = systems.water_box(20, seed=1, rigid=True); dt = 0.002; cfg = MdConfig(coulomb_mode=1)
else:
    s = systems.BY_NAME[name](); dt = 0.0005; cfg = MdConfig(coulomb_mode=1)       # reaction field: continuous energy
with MdState(s, cfg) as md:
    md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=5)
    md.set_thermostat(1, 300.0, 0.02, 1); md.step(dt, None, 600); md.set_thermostat(0, 300.0, 0.02, 1)
    if name == "rigid":
        md.set_thermostat(2, 300.0, 0.1, 10, seed=2); md.set_barostat(1, 1.0, 1.0, 4.5e-5, 25)
    e0 = md.energy(); t0 = time.time()
    print("start T %.1f Etot %.2f P %.0f bar V %.0f" % (e0["temperature"], e0["potential"] + e0["kinetic"], e0["pressure"], e0["volume"]))
    n = steps // 10
    for k in range(10):
        md.step(dt, None, n)
        e = md.energy(); st = md.stats()
        print("%6d steps: T %.1f  dE/atom %.5f  maxF %.0f  P %.0f bar  density %.4f  rebuilds %d" % ((k + 1) * n, e["temperature"], (e["potential"] + e["kinetic"] - e0["potential"] - e0["kinetic"]) / s.n_atoms, e["max_force"], e["pressure"], e["density"] * 1.66054, st["rebuild_count"]), flush=True)
    print("wall %.1f s" % (time.time() - t0))
