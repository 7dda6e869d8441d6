"""Throughput at the reference's DEFAULT operating point (/root/reference src/prefs/mod.rs:203, src/ui/panels/md.rs:362-371,
README.md:236-240): rigid 4-site OPC water (SHAKE/RATTLE + M virtual site), dt = 2 fs, SPME, CSVR thermostat - on a
1,048,576-site box (64^3 waters), next to the flexible cutoff box the headline metric is quoted on."""
import os, sys, time
if os.environ.get("DP_IMPORT_TORCH") == "1": import torch      # (A/B: bench.py's process has PyTorch's bundled HIP runtime loaded)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inner = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0      # dual-list buffer (0 = library default 0.5 A)
skin = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 300         # untimed steps at the operating point (the library's dual-list tuning settles within ~1500)
s = systems.opc_water_box(n_side, seed=5)
cases = (("SPME", MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0, inner_skin=inner, skin=skin)), ("cutoff (reaction field)", MdConfig(coulomb_mode=1, inner_skin=inner, skin=skin)))
if os.environ.get("DP_ONLY") == "spme": cases = cases[:1]
if os.environ.get("DP_ONLY") == "rf": cases = cases[1:]
for name, cfg in cases:
    with MdState(s, cfg) as md:
        md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.001, None, 1500)       # untimed: the random-orientation lattice relaxes
        md.set_thermostat(2, 300.0, 0.1, 10, seed=2); md.step(0.002, None, pre)
        n = 500 if s.n_atoms >= 500000 else 5000      # (500 steps of a 23 k-site box are 60 ms: inside the GPU's return from idle, profiles/r05_fresh_handle_ramp.txt)
        t = time.perf_counter(); md.step(0.002, None, n); e = md.energy(); el = time.perf_counter() - t
        st = md.stats()
        if skin == 0.0:
            print("   library-chosen skin: %.2f A (still tuning: %s)" % md.skin(), flush=True)
        print("inner_skin %.1f skin %.1f | OPC %d sites, dt 2 fs, %s: %.0f steps/s = %.1f ns/day  (T %.0f K, %d list rebuilds, %d pruning passes, inner/verlet cluster pairs %.2f, %d rebuilds left the fused chain)" % (
            inner or 0.5, skin, s.n_atoms, name, n / el, n / el * 0.002e-3 * 86400, e["temperature"], st["rebuild_count"], st["prune_passes"],
            st["n_inner_cluster_pairs"] / max(st["n_cluster_pairs"], 1), st.get("rebuild_fallbacks", 0)), flush=True)
