import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
name = sys.argv[1] if len(sys.argv) > 1 else "dhfr23k"
s = systems.BY_NAME[name]()
md = MdState(s, MdConfig())
for k in range(30):
    e = md.energy()
    print(k * 100, "T %.0f pot %.1f kin %.1f tot %.1f maxF %.1f bond %.1f lj %.1f coul %.1f" % (e["temperature"], e["potential"], e["kinetic"], e["potential"] + e["kinetic"], e["max_force"], e["bond"], e["lj"], e["coulomb"]), flush=True)
    try:
        md.step(0.0005, None, 100)
    except Exception as ex:
        print("FAILED", ex)
        f = md.forces(); v = md.velocities(); p = md.positions()
        i = int(np.argmax(np.abs(v).max(1))); print("fastest atom", i, v[i], p[i], "n_chain 2489")
        break
