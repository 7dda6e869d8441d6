import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
s = systems.dhfr23k()
md = MdState(s, MdConfig())
md.step(0.0005, None, 100)
L = 62.23
for k in range(100):
    f = md.forces(); p = md.positions(); v = md.velocities()
    fm = np.linalg.norm(f, axis=1); i = int(np.argmax(fm))
    if fm[i] > 1500 or k % 20 == 0:
        d = p - p[i]; d -= np.round(d / L) * L; r = np.linalg.norm(d, axis=1); r[i] = 9
        j = np.argsort(r)[:4]
        print(k, "maxF %.0f atom %d type %d q %.2f | nearest:" % (fm[i], i, s.lj_type[i], s.charge[i]),
              [(int(a), round(float(r[a]), 2), int(s.lj_type[a]), round(float(s.charge[a]), 2)) for a in j], "vmax %.1f" % np.abs(v).max(), flush=True)
    if fm[i] > 1e5: break
    md.step(0.0005, None, 1)
