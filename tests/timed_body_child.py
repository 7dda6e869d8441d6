"""Child of tests/test_gpu_timed_body.py: one process per MDX_WPT value (the library reads the knob once).

usage: MDX_WPT=<w> python tests/timed_body_child.py <w> [fused]
Steps two small systems through the step loop with <w> waves per tile and holds the forces the step loop left behind (merged
dual-list body) against the oracle, at several points of the trajectory (before / after pruning passes and a list rebuild).
`fused` (with MDX_WPT8_BELOW=32 in the environment) also takes the large classes' fused bonded + kick + drift pass, i.e. the
whole arrangement bench.py times at water1M, on a 12 k-atom box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from molchanica_amd import MdConfig, systems  # noqa: E402
from molchanica_amd import md_state  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_gpu_timed_body import step_loop_forces_vs_oracle  # noqa: E402


def main():
    w = int(sys.argv[1])
    fused = len(sys.argv) > 2 and sys.argv[2] in ("fused", "inner", "onepass", "generic")
    inner = len(sys.argv) > 2 and sys.argv[2] == "inner"
    assert os.environ.get("MDX_WPT") == str(w)
    assert md_state.device_count() >= 1
    orc.lib()
    cases = [
        ("water12k", systems.water_box(16, seed=41), MdConfig()),                                   # the bench's cutoffs and Coulomb mode
        ("chain-in-water", systems.small_solvated(n_chain=400, box=44.0), MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0)),
    ]
    for name, s, cfg in cases:
        with md_state.MdState(s, cfg) as md:
            if name != "water12k":
                md.minimize_energy(40)
                md.initialize_velocities(400.0, True, seed=3)
            if fused and os.environ.get("MDX_ONEPASS", "1") != "2":
                md.profile(1)       # (event brackets count the fused pass: mdx_stats.fused_launches; they also keep the separate passes)
            done = 0
            for burst in (7, 20, 33):
                md.step(0.0005, None, burst)
                done += burst
                info = md.pair_launch_info()["step"]
                assert info["waves_per_tile"] == w and info["half"] == 1 and info["energy"] == 0, info
                # w < 8: the merged launch (3); w = 8 below 1024 tiles: two workgroups per tile, the bonded gather riding along (4)
                # round 6, one launch per step: w = 8 by default (6), w = 1 with the fused drift pass selected under MDX_ONEPASS=2 (5)
                op = os.environ.get("MDX_ONEPASS", "1")
                onepass = (w == 8 and op != "0") or (fused and w == 1 and name == "water12k" and op == "2")
                # (the force call behind a list rebuild is body 3 / 4 there too: the launch counter speaks for the steps in between)
                assert info["dual"] in (((6, 4) if w == 8 else (5, 3)) if onepass else ((3,) if w != 8 else (4,))), info
                assert (md.pair_launch_info()["one_launch_steps"] > 0) == onepass, md.pair_launch_info()
                step_loop_forces_vs_oracle(md, orc, s, cfg, f"{name} wpt {w} after {done} steps")
            st = md.stats()
            assert st["prune_passes"] >= 3 and st["rebuild_count"] >= 2, (st["prune_passes"], st["rebuild_count"])
            if inner:
                n_in = md.pair_launch_info()["inner_lists_from_rebuilds"]
                assert n_in >= 1, (n_in, st["rebuild_count"])      # (every rebuild of the fused chain; a handle's first takes the unfused one)
            if fused and name == "water12k" and not onepass:      # (the chain's mean role count keeps it on the separate bonded gather: mdx_bonded_integrate_ok)
                assert st["fused_launches"] > 0, "the fused bonded + kick + drift pass did not run"
            print(f"{name}: wpt {w} dual {info['dual']} tiles {info['tiles']} prune passes {st['prune_passes']} rebuilds {st['rebuild_count']}"
                  f" fused launches {st['fused_launches']}")
    print("TIMED-BODY-OK")


if __name__ == "__main__":
    main()
