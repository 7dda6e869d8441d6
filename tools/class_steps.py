"""A short run of one of the small classes without event brackets (for rocprofv3 --kernel-trace: the arrangement the driver line's `classes`
block times).  python tools/class_steps.py dhfr23k [steps=600]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from molchanica_amd import MdConfig, systems, md_state
name = sys.argv[1] if len(sys.argv) > 1 else "dhfr23k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
s = systems.BY_NAME[name]()
cfg = MdConfig()
with md_state.MdState(s, cfg) as eq:
    eq.minimize_energy(100); eq.initialize_velocities(300.0, True, seed=105)
    eq.set_thermostat(1, 300.0, 0.02, 1); eq.step(0.0005, None, 600); eq.set_thermostat(0, 300.0, 0.02, 1)
    s.pos, s.vel = np.ascontiguousarray(eq.positions(), np.float32), np.ascontiguousarray(eq.velocities(), np.float32)
with md_state.MdState(s, cfg) as md:
    md.step(0.0005, None, n)
    print(md.pair_launch_info())
