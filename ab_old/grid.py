import sys, os, time
sys.path.insert(0, os.getcwd())
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
s = systems.opc_water_box(64, seed=5)
for g in (0, 192, 216, 240, 256):
    cfg = MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0, pme_grid=(g, g, g) if g else (0, 0, 0))
    with MdState(s, cfg) as md:
        md.minimize_energy(50); md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.001, None, 300)
        md.set_thermostat(2, 300.0, 0.1, 10, seed=2); md.step(0.002, None, 100)
        t = time.perf_counter(); md.step(0.002, None, 300); e = md.energy(); el = time.perf_counter() - t
        print(f"grid {g or 'auto'}: {300 / el:.0f} steps/s  recip {e['coulomb_recip']:.1f}", flush=True)
