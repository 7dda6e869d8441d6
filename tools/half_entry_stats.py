"""How much would finer evaluation units buy the pair kernel?  Counts, inside the pruning pass, the j-halves / i-halves /
quadrants of every kept cluster pair that hold an atom pair within cutoff + inner_skin.  Needs the library built with the
counters compiled in:  make -C molchanica_amd/csrc clean all EXTRA=-DNB_HALF_STATS   (result in DESIGN.md section 4)."""
import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState, load_library
lib = load_library()
lib.mdx_debug_half_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for name, s in (("water 40^3", systems.water_box(40)), ("dhfr23k", systems.dhfr23k())):
    with MdState(s, MdConfig()) as md:
        md.minimize_energy(50); md.initialize_velocities(300.0, True, seed=1)
        md.step(0.0005, None, 200)
        out = (C.c_ulonglong * 6)()
        lib.mdx_debug_half_stats(md._h, out)
        passes, jh, ih, q, kept, pairs = [int(v) for v in out]
        print(f"{name}: passes {passes} kept cluster pairs/pass {kept/passes:.0f}: j-halves {jh/kept:.3f}/2  i-halves {ih/kept:.3f}/2  quadrants {q/kept:.3f}/4  "
              f"in-range lane pairs per kept cluster pair {pairs/kept:.1f}/64 | evals if j-half units: {jh/kept/2:.3f}x, quadrant units: {q/kept/4:.3f}x")
