#!/usr/bin/env python3
"""What would a split pair loop cost?  (round 3 lead: Coulomb-only pass over all atoms + LJ-only pass over the LJ atoms)

    python3 tools/exp_split_cost.py prep            # normal library: equilibrate water1M, save the state
    python3 tools/exp_split_cost.py time TAG        # whatever MDX_LIB points at: inner-walk kernel time on that state (atoms frozen: dt ~ 0)
    python3 tools/exp_split_cost.py oxygens TAG     # the 343 k oxygens alone, LJ only (q = 0), same measurement

The atoms do not move between the force calls (dt = 1e-9 ps), so after the first pruning pass every launch is an
inner-list walk of the same list: the kernel time is that of the steady state without its pruning passes.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from molchanica_amd import MdConfig, systems  # noqa: E402
from molchanica_amd._abi import MdSystem  # noqa: E402
from molchanica_amd.md_state import MdState  # noqa: E402

STATE = os.path.join("/tmp", "exp_state_water1M.npz")


def measure(system, cfg, tag, n=60):
    with MdState(system, cfg) as md:
        md.step(1e-9, None, 20)
        md.profile(2)
        s0 = md.stats()
        md.step(1e-9, None, n)
        s1 = md.stats()
        out = {"tag": tag, "n_atoms": system.n_atoms,
               "nb_ms": (s1["nb_ms_sum"] - s0["nb_ms_sum"]) / max(s1["nb_launches"] - s0["nb_launches"], 1),
               "launches": s1["nb_launches"] - s0["nb_launches"],
               "prune_passes": s1.get("prune_passes", 0) - s0.get("prune_passes", 0),
               "inner_cluster_pairs": s1.get("n_inner_cluster_pairs"), "cluster_pairs": s1.get("n_cluster_pairs"),
               "lib": os.environ.get("MDX_LIB", "default")}
        print(json.dumps(out), flush=True)


def main():
    mode = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else mode
    cfg = MdConfig(skin=2.0, chunk_steps=16)
    if mode == "prep":
        system = systems.BY_NAME["water1M"]()
        with MdState(system, cfg) as eq:
            eq.minimize_energy(100)
            eq.initialize_velocities(300.0, True, seed=105)
            eq.set_thermostat(1, 300.0, 0.02, 1)
            eq.step(0.0005, None, 600)
            pos, vel = eq.positions(), eq.velocities()
        os.makedirs(os.path.dirname(STATE), exist_ok=True)
        np.savez(STATE, pos=pos, vel=vel)
        return
    st = np.load(STATE)
    system = systems.BY_NAME["water1M"]()
    system.pos = np.ascontiguousarray(st["pos"], np.float32)
    system.vel = np.zeros_like(system.pos)
    if mode == "time":
        measure(system, cfg, tag)
    elif mode == "oxygens":
        o = np.ascontiguousarray(system.pos[0::3])
        n = o.shape[0]
        sub = MdSystem(pos=o, mass=np.full(n, 15.9994, np.float32), charge=np.zeros(n, np.float32),
                       lj_type=np.zeros(n, np.uint32), lj_sigma=[systems.TIP3P["o_sigma"]], lj_eps=[systems.TIP3P["o_eps"]],
                       vel=np.zeros((n, 3), np.float32), periodic=True, box_lo=system.box_lo, box_hi=system.box_hi,
                       name="oxygens").normalise()
        measure(sub, cfg, tag)


if __name__ == "__main__":
    main()
