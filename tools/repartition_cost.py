"""What a REPARTITION costs one rank of N (null transport, production path): rank 0 of the N-rank decomposition of the 1 M-atom
box, every stale list forced to be a repartition; prints their wall time (gather kernels + classify + scan + fill; the wire time
of the all-gather is not in it).  In a real run a repartition replaces about every third list rebuild (profiles/r02_decomp_soak_water1M.txt:
4 repartitions + 8 local rebuilds in 300 steps).
Usage: python tools/repartition_cost.py [world=8] [steps=48]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
# A lone rank gets no ghost forces back and its box is unphysical after ~70 steps, so the run is short and every stale list is
# MADE a repartition: with a halo margin of 0.2 A "no atom has drifted further than margin / 2 since the last partition" never holds
os.environ["MDX_HALO_MARGIN"] = os.environ.get("MDX_HALO_MARGIN", "0.2")
s = systems.water1m()
with MdState(s, MdConfig()) as eq:
    eq.minimize_energy(100); eq.initialize_velocities(300.0, True, seed=105)
    eq.set_thermostat(1, 300.0, 0.02, 1); eq.step(0.0005, None, 600); eq.set_thermostat(0, 300.0, 0.02, 1)
    s.pos = np.ascontiguousarray(eq.positions(), dtype=np.float32); s.vel = np.ascontiguousarray(eq.velocities(), dtype=np.float32)
os.environ["MDX_HALO_OVERLAP"] = "0"
with MdState(s, MdConfig()) as md:
    md.comm_init_null(0, world)
    md.step(0.0005, None, 8)
    st0 = md.stats(); t0 = time.perf_counter()
    md.step(0.0005, None, steps)
    st1 = md.stats(); el = time.perf_counter() - t0
    rp = st1["repartitions"] - st0["repartitions"]; lr = st1["local_rebuilds"] - st0["local_rebuilds"]
    rp_ms = st1["repartition_ms_sum"] - st0["repartition_ms_sum"]
    print("world %d rank 0: %d steps in %.1f ms = %.3f ms per step; %d repartitions (%.2f ms each incl. gather, classify, scan, fill; "
          "their list rebuild comes on top), %d local rebuilds -> at one repartition per 75 steps that is %.4f ms per step = %.1f %% of this rank's step" % (
              world, steps, 1e3 * el, 1e3 * el / steps, rp, rp_ms / max(rp, 1), lr, rp_ms / max(rp, 1) / 75.0, 100 * (rp_ms / max(rp, 1) / 75.0) / (1e3 * el / steps)), flush=True)
