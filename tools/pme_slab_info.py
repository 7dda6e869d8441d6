"""Mesh traffic of the slab-decomposed SPME at the headline size: rank 0 of WORLD on the 1 M-site OPC box (240^3 mesh), null transport.
Usage: python tools/pme_slab_info.py [world=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = systems.water1m()
cfg = MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0)
with MdState(s, cfg) as md:
    md.comm_init_null(0, world)
    i = md.pme_info(); c = md.comm_info()
    print("world %d rank 0: mesh %s, slab %s: sends %.2f MB of real-space mesh (charges out + potential back) + %.2f MB of FFT transposes per force call; "
          "a replicated mesh is %.1f MB (all-reduced: in and out of every rank); owned %d ghosts %d" % (
              world, "x".join(str(k) for k in (240, 240, 240)), i["slab_on"], i["mesh_bytes_sent"] / 1e6, i["transpose_bytes_sent"] / 1e6,
              i["replicated_mesh_bytes"] / 1e6, c["n_owned"], c["n_ghost"]))
