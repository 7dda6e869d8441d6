"""Does an MI355X that sat idle come back at full speed?  One water1M handle: 400 steps, then for each idle time (host sleep) 40
steps - under rocprofv3 --kernel-trace the pair-kernel durations right behind each pause show the ramp (tools/idle_ramp.py prints
them itself from HIP-side timing when run without the profiler: wall time of the first 8 steps and of steps 33-40 behind a pause).
Usage (through gpurun): python3 tools/idle_ramp.py [water1M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
name = sys.argv[1] if len(sys.argv) > 1 else "water1M"
s = systems.BY_NAME[name]()
with MdState(s, MdConfig()) as md:
    md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=1)
    md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.0005, None, 400); md.set_thermostat(0, 300.0, 0.02, 1)
    md.step(0.0005, None, 400); md.stats()
    for idle_ms in (0, 1, 5, 20, 50, 100, 300, 1000):
        time.sleep(idle_ms * 1e-3)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); md.step(0.0005, None, 8); md.stats(); t.append((time.perf_counter() - t0) / 8 * 1e3)
        print(f"{name}: idle {idle_ms:5d} ms -> ms per step over steps 1-8: {t[0]:.4f} | 9-16: {t[1]:.4f} | 17-24: {t[2]:.4f} | 25-32: {t[3]:.4f} | 33-40: {t[4]:.4f}", flush=True)
        md.step(0.0005, None, 200); md.stats()
