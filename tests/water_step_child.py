"""Child of tests/test_gpu_constraints.py::test_one_pass_water_step_against_the_three_launches: MDX_WATER_STEP is read once per process.
usage: MDX_WATER_STEP=<0|1> python tests/water_step_child.py <out.npz>
Rigid TIP3P and OPC boxes (lattice start and one with waters straddling the box faces), 60 steps at dt 2 fs in bursts that put chunk
boundaries, list rebuilds and pruning passes at different places, with and without the SPME mesh: positions, velocities, energies,
virial / pressure of every case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from molchanica_amd import MdConfig, systems  # noqa: E402
from molchanica_amd.md_state import MdState  # noqa: E402


def main():
    out = {}
    cases = []
    for model in ("tip3p_rigid", "opc"):
        s = systems.water_box(8, seed=3, rigid=True) if model == "tip3p_rigid" else systems.opc_water_box(8, seed=3)
        cases.append((model, s, MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)))
    s = systems.opc_water_box(8, seed=7)
    L = np.array(s.box_hi, dtype=np.float64)
    s.pos = np.mod(np.asarray(s.pos, dtype=np.float64) + 1.25, L).astype(np.float32)      # waters straddle every upper face
    cases.append(("opc_straddling_spme", s, MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=2, ewald_alpha=0.35, overrides=0)))
    for name, s, cfg in cases:
        with MdState(s, cfg) as md:
            md.minimize_energy(60)
            md.initialize_velocities(300.0, True, seed=2)
            for burst in (1, 7, 16, 3, 33):
                md.step(0.002, None, burst)
            e = md.energy()
            st = md.stats()
            out[name + "_pos"] = md.positions(); out[name + "_vel"] = md.velocities(); out[name + "_frc"] = md.forces()
            out[name + "_e"] = np.array([e[k] for k in ("potential", "kinetic", "virial", "pressure", "coulomb_recip")])
            out[name + "_rebuilds"] = np.array([st["rebuild_count"], st["prune_passes"]])
    np.savez(sys.argv[1], **out)
    print("WATER-STEP-CHILD-OK")


if __name__ == "__main__":
    main()
