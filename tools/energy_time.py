"""Times the energy-flavoured force pass (what snapshots, the minimiser and the barostat pay)."""
import sys, time
sys.path.insert(0, ".")
from molchanica_amd import md_state as mdx, systems, MdConfig
s = systems.water1m()
with mdx.MdState(s, MdConfig()) as md:
    md.step(0.0005, None, 20)
    e = md.energy()
    t = time.perf_counter()
    for _ in range(20): e = md.energy()
    print("energy call: %.3f ms  (pot %.3f vir %.3f P %.1f bar)" % ((time.perf_counter() - t) / 20 * 1e3, e["potential"], e["virial"], e["pressure"]))
