"""Null transport + pinned half shell + stated wire time on memory a previous handle has used (tests/test_gpu_comm.py: the shell-choice test in suite order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState, compute_energy_snapshot
CFG = dict(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, chunk_steps=8)
# dirty the allocator: handles of other systems come and go, the scorer keeps one
with MdState(systems.dhfr23k(), MdConfig()) as md: md.step(0.0005, None, 40)
with MdState(systems.opc_water_box(10, seed=5), MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0)) as md: md.step(0.002, None, 40)
compute_energy_snapshot(systems.small_solvated(), MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0))
for wire, pin in (("0", None), ("25", None), ("25", "1"), ("25", "1"), ("0", "1")):
    os.environ["MDX_NULL_WIRE_US"] = wire
    if pin is None: os.environ.pop("MDX_HALF_SHELL", None)
    else: os.environ["MDX_HALF_SHELL"] = pin
    s = systems.water_box(14, seed=6)
    try:
        with MdState(s, MdConfig(**CFG)) as md:
            md.comm_init_null(0, 8)
            d = md.comm_diag()
            for k in range(6):
                md.step(0.0005, None, 1)
                st = md.stats()
                print(f"wire {wire} pin {pin}: step {k + 1}: half_shell {d['half_shell']} rebuilds {st['rebuild_count']} local rebuilds {st['local_rebuilds']} repartitions {st['repartitions']}", flush=True)
    except Exception as e:
        print(f"wire {wire} pin {pin}: FAILED {str(e)[:200]}", flush=True)
