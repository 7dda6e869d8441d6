"""Child of tests/test_gpu_onepass.py: one launch per step with the words of every launch UNDER-estimating the next stage
(MDX_ONEPASS_GRANT < 0, read once per process), so that launches find themselves contradicted - walking the inner list with the path budget
spent, or running on a stale list - and the host takes those steps back to a list rebuild.  The trajectory must still be the oracle's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from molchanica_amd import MdConfig, systems  # noqa: E402
from molchanica_amd import md_state  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_gpu_timed_body import step_loop_forces_vs_oracle  # noqa: E402


def main():
    assert float(os.environ["MDX_ONEPASS_GRANT"]) < 0
    s = systems.small_solvated(n_chain=400, box=44.0)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0)
    with md_state.MdState(s, cfg) as md:
        md.minimize_energy(40)
        md.initialize_velocities(400.0, True, seed=3)
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        md.step(0.0005, None, 60)
        info, st = md.pair_launch_info(), md.stats()
        assert info["one_launch_steps"] > 30 and info["kicks_beyond_grant"] >= 2, info
        assert st["rebuild_count"] >= info["kicks_beyond_grant"] + 1
        x = md.positions().astype(np.float64)
        step_loop_forces_vs_oracle(md, orc, s, cfg, "after steps taken back")
    xo, _, _ = orc.step(s, cfg, 0.0005, 60, pos=x0, vel=v0, use_cells=True)
    L = np.array(s.box_hi, np.float64) - np.array(s.box_lo, np.float64)
    d = x - xo
    d -= np.round(d / L) * L
    rms = float(np.sqrt((d ** 2).sum(1).mean()))
    assert rms < 1e-3, rms
    print(f"steps taken back: {info['kicks_beyond_grant']} of {info['one_launch_steps']} launches, {st['rebuild_count']} rebuilds, rms against the oracle {rms:.2e} A")
    print("ONEPASS-CHILD-OK")


if __name__ == "__main__":
    main()
