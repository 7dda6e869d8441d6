"""The GUI drives the engine in bursts (`run_dynamics_blocking`: 10 steps per frame, /root/reference src/md/mod.rs:729-750):
throughput of repeated mdx_step(h, dt, NULL, 10) calls against one long call.  Usage: python tools/burst_time.py [dhfr23k|opc18]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
for name in (sys.argv[1:] or ["dhfr23k", "opc18"]):
    if name == "opc18":
        s = systems.opc_water_box(18, seed=5); cfg = MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0); dt = 0.002
    else:
        s = systems.BY_NAME[name](); cfg = MdConfig(); dt = 0.0005
    with MdState(s, cfg) as md:
        md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.02, 1); md.step(min(dt, 0.001), None, 1000); md.set_thermostat(0, 300.0, 0.02, 1)
        md.step(dt, None, 500)
        t = time.perf_counter(); md.step(dt, None, 3000); md.stats(); long_rate = 3000 / (time.perf_counter() - t)
        out = []
        for burst in (10, 25, 100):
            n = 3000 // burst
            t = time.perf_counter()
            for _ in range(n): md.step(dt, None, burst)
            md.stats(); out.append((burst, n * burst / (time.perf_counter() - t)))
        print(f"{name}: one call of 3000 steps {long_rate:.0f} steps/s | " + " | ".join(f"bursts of {b}: {r:.0f} steps/s" for b, r in out), flush=True)
