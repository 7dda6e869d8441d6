"""The reference's default operating point (rigid 4-site OPC water, dt 2 fs, CSVR, SPME: README.md:236-240) as ONE rank of an N-rank
decomposition sees it: rank 0 of WORLD on the 1,048,576-site box, alone on one MI355X (null transport: kernel and host costs of a
real rank, no wire time; short fresh-handle stretches, see tools/one_rank_profile.py), slab-decomposed mesh against the replicated
one (MDX_PME_SLAB=0).  Usage: python tools/default_point_one_rank.py [world=8] [n_side=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
s = systems.opc_water_box(n_side, seed=5)
cfg = MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0)
with MdState(s, cfg) as md:      # untimed preparation on the whole box
    md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=1)
    md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.001, None, 1500)
    md.set_thermostat(2, 300.0, 0.1, 10, seed=2); md.step(0.002, None, 200)
    s.pos = np.ascontiguousarray(md.positions(), dtype=np.float32); s.vel = np.ascontiguousarray(md.velocities(), dtype=np.float32)
    t = time.perf_counter(); md.step(0.002, None, 100); md.stats(); one = (time.perf_counter() - t) / 100
print("one GPU, whole box (%d sites): %.3f ms per step" % (s.n_atoms, 1e3 * one), flush=True)
for slab in ("1", "0"):
    os.environ["MDX_PME_SLAB"] = slab
    wall = 0.0; n = 0
    for rep in range(4):
        with MdState(s, cfg) as md:
            md.set_thermostat(2, 300.0, 0.1, 10, seed=2)
            md.comm_init_null(0, world)
            md.step(0.002, None, 8)
            md.stats()
            t = time.perf_counter(); md.step(0.002, None, 40); md.stats(); wall += time.perf_counter() - t; n += 40
            info = md.pme_info(); c = md.comm_info()
    print("rank 0 of %d, %s mesh: %.3f ms per step (owned %d, ghosts %d; sends %.1f + %.1f MB of mesh per force call) -> ceiling %.0f steps/s = %.0f ns/day without wire time" % (
        world, "slab-decomposed" if info["slab_on"] else "replicated (all-reduce a no-op here)", 1e3 * wall / n, c["n_owned"], c["n_ghost"],
        info["mesh_bytes_sent"] / 1e6, info["transpose_bytes_sent"] / 1e6, n / wall, n / wall * 0.002e-3 * 86400), flush=True)
os.environ.pop("MDX_PME_SLAB", None)
