"""A/B of the full-list (variant 2) and half-list (variant 5) pair kernels: parity of forces,
energies and neighbour lists between the two, then HIP-event kernel times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState

names = sys.argv[1:] or ["small", "dhfr23k", "dna100k", "water1M"]
for name in names:
    s = systems.small_solvated() if name == "small" else systems.BY_NAME[name]()
    kw = {} if s.periodic else dict(lj_cutoff=0.0, coulomb_cutoff=0.0)
    ref = None
    for variant in (2, 5, 2, 5):
        md = MdState(s, MdConfig(nb_variant=variant, **kw))
        f = md.forces().astype(np.float64); e = md.energy()
        if s.n_atoms < 100000:
            off, idx = md.neighbor_list()
        else:
            off, idx = np.zeros(1), np.zeros(1)
        if ref is None: ref = (f, e, off, idx)
        else:
            frms = np.sqrt((ref[0] ** 2).sum(1).mean())
            print("   v%d vs v2: max|dF| %.3e (rms F %.2f)  dE_lj %.3e dE_coul %.3e  nl equal %s  sumF %s" % (
                variant, np.abs(f - ref[0]).max(), frms, e["lj"] - ref[1]["lj"], e["coulomb"] - ref[1]["coulomb"],
                np.array_equal(off, ref[2]) and np.array_equal(idx, ref[3]), np.abs(f.sum(0)).max()))
        md.step(0.0005, None, 20)
        md.profile(True)
        t = time.time(); md.step(0.0005, None, 200); dt = time.time() - t
        st = md.stats()
        print("%s v%d: %.1f steps/s  nb %.3f ms (%d)  bonded %.3f  integ %.3f  rebuilds %d @ %.2f ms  entries %d cluster_pairs %d" % (
            name, variant, 200 / dt, st["nb_ms_sum"] / max(st["nb_launches"], 1), st["nb_launches"],
            st["bonded_ms_sum"] / max(st["bonded_launches"], 1), st["integ_ms_sum"] / max(st["integ_launches"], 1),
            st["rebuild_count"], st["rebuild_ms_sum"] / max(st["rebuild_count"] - 1, 1), st["n_list_entries"], st["n_cluster_pairs"]))
        e2 = md.energy()
        print("      E_tot after 220 steps: %.4f (start %.4f)" % (e2["potential"] + e2["kinetic"], e["potential"] + e["kinetic"]))
        md.close()
