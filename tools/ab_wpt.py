"""Pair-kernel time vs waves per tile (MDX_WPT) over system sizes: run once per MDX_WPT value."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
for n in [int(x) for x in sys.argv[1:]] or [30, 40, 50, 60, 70]:
    s = systems.water_box(n)
    with MdState(s, MdConfig()) as md:
        md.minimize_energy(30); md.initialize_velocities(300.0, True, seed=1)
        md.step(0.0005, None, 60)
        md.profile(2); md.step(0.0005, None, 200)
        st = md.stats()
        print("wpt %s  atoms %7d tiles %6d  nb %.4f ms" % (os.environ.get("MDX_WPT", "auto"), s.n_atoms, st["n_tiles"], st["nb_ms_sum"] / max(st["nb_launches"], 1)), flush=True)
