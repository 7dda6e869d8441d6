"""One launch per step (round 6): below 2048 tiles the pair launch of the step loop also finishes the previous step for its tile's atoms -
kick, drift, path lengths, the words that gate the next launch - and the bonded gather rides in its extra workgroups; positions travel
in a "step form" Y = x + dt v between two buffers, forces rotate through three (csrc/mdx_nonbonded_impl.h STEP, mdx_api.hip mdx_step).
`MdState::step` is the path (/root/reference src/md/mod.rs:716,748).  The oracle comparisons of the whole suite run through this
arrangement by default (every small system is in the eight-waves class); here: the arrangement against the separate passes it replaces
for every shape of a step call, the chunks that end with an energy evaluation, a stale list inside such a chunk, and the way back when a
launch finds its gating words contradicted."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1, "no GPU: the HIP path must run here, there is no fallback"
    return md_state


@pytest.mark.parametrize("cadence,calls", [(0, (30,)), (0, (10, 10, 10)), (10, (30,)), (10, (10, 10, 10)), (10, (4, 6, 10, 3, 7)), (0, (1,) * 12), (7, (1,) * 15)])
def test_one_launch_per_step_against_the_separate_passes(mdx, cadence, calls, monkeypatch):
    """MDX_ONEPASS is read per chunk: both arms in one process.  The two arrangements round differently (x + dt v + w F in two steps
    against x + dt (v + kdt F)), so they part like any two runs: measured 3e-6 ... 6e-6 A rms after 30 steps, velocities 1e-3 A/ps.
    cadence 10 with a 30-step call: the second chunk's list goes stale INSIDE a chunk that ends with an energy evaluation (the pair launch
    of that evaluation enqueues an ungated fill - it once wiped the force rows the way back out of the step form reads)."""
    s = systems.small_solvated()
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    L = np.array(s.box_hi) - np.array(s.box_lo)
    out = {}
    for arm in ("1", "0"):
        monkeypatch.setenv("MDX_ONEPASS", arm)
        with mdx.MdState(s, cfg) as md:
            if cadence:
                md.set_snapshot_cadence(cadence, with_velocities=True)
            rows = []
            for n in calls:
                md.step(0.0005, None, n)
                rows.append((md.positions().astype(np.float64), md.velocities().astype(np.float64), md.energy()))
            info, st = md.pair_launch_info(), md.stats()
            out[arm] = (rows, info, st, [sn["step"] for sn in md.snapshots] if cadence else [])
    assert out["1"][1]["one_launch_steps"] > 0 and out["0"][1]["one_launch_steps"] == 0, (out["1"][1], out["0"][1])
    assert out["1"][1]["kicks_beyond_grant"] == 0
    assert out["1"][3] == out["0"][3]
    assert abs(out["1"][2]["rebuild_count"] - out["0"][2]["rebuild_count"]) <= 1
    for (xa, va, ea), (xb, vb, eb) in zip(out["1"][0], out["0"][0]):
        d = xa - xb
        d -= np.round(d / L) * L
        assert math.sqrt((d ** 2).sum(1).mean()) < 5e-5 and math.sqrt(((va - vb) ** 2).sum(1).mean()) < 1e-2
        for k in ("bond", "angle", "dihedral", "lj", "coulomb", "kinetic"):
            assert ea[k] == pytest.approx(eb[k], rel=2e-5, abs=0.05), k


def test_launches_that_contradict_their_words_are_taken_back():
    env = {k: v for k, v in os.environ.items() if not k.startswith("MDX_")}
    env["MDX_ONEPASS_GRANT"] = "-40"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "onepass_child.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-25:])
    assert p.returncode == 0 and "ONEPASS-CHILD-OK" in p.stdout, tail


def test_nan_coordinates_stop_the_step_loop_in_this_arrangement_too(mdx):
    """A runaway / non-finite coordinate raises the stale word beyond 1e29 whichever pass finds it (mdx_step: MDX_ENAN)."""
    s = systems.water_box(6, seed=2)
    with mdx.MdState(s, MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0)) as md:
        md.step(0.0005, None, 5)
        assert md.pair_launch_info()["one_launch_steps"] > 0
        v = md.velocities()
        v[7] = np.float32(1.0e19)        # (uploads refuse inf / NaN; this one runs away within a step)
        md.set_velocities(v)
        with pytest.raises(Exception, match="non-finite|runaway"):
            md.step(0.0005, None, 20)
