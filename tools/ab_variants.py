"""A/B of the pair-kernel variants on water boxes (HIP-event kernel times)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState

sizes = [int(x) for x in sys.argv[1:]] or [40, 70]
for n in sizes:
    s3 = systems.water_box(n)
    f_ref = None
    for variant in (4, 2, 4, 2):
        md3 = MdState(s3, MdConfig(nb_variant=variant))
        f = md3.forces()
        if f_ref is None: f_ref = f
        else: print("   max |dF| vs first variant: %.3e" % np.abs(f - f_ref).max())
        md3.step(0.0005, None, 20)
        md3.profile(True)
        t = time.time(); md3.step(0.0005, None, 100); dt = time.time() - t
        st = md3.stats()
        print("water %d v%d: %.1f steps/s  nb %.3f ms (%d)  bonded %.3f ms  integ %.3f ms rebuilds %d rebuild_ms/each %.2f  entries %d cluster_pairs %d (%.0f evals/atom)" % (
            s3.n_atoms, variant, 100 / dt, st["nb_ms_sum"] / max(st["nb_launches"], 1), st["nb_launches"], st["bonded_ms_sum"] / max(st["bonded_launches"], 1),
            st["integ_ms_sum"] / max(st["integ_launches"], 1), st["rebuild_count"], st["rebuild_ms_sum"]/max(st["rebuild_count"]-1,1), st["n_list_entries"], st["n_cluster_pairs"], st["n_cluster_pairs"]*64/s3.n_atoms))
        md3.close()
