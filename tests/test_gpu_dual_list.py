"""Dual pair list (`mdx_config.inner_skin`): the step loop's pair kernel walks a rolling-pruned inner list.

The inner list is an optimisation of the Verlet list the reference's engine keeps behind `MdState::step`
(/root/reference src/md/mod.rs:716,748; neighbour refresh inferred): it must never change a result.
Checked here on the GPU:
  * the forces the step loop left behind (evaluated over the INNER masks) equal a fresh evaluation of the same
    positions over the plain list (`mdx_energy` walks the outer masks) at many points of a hot trajectory, i.e.
    also on the last steps before a re-pruning pass,
  * trajectories with the dual list on and off agree,
  * the pruning passes happen on the device (statistics), keep fewer cluster pairs than the Verlet list, and
  * SHAKE corrections and virtual sites (rigid TIP3P, OPC) are covered by the same path-length bound.
"""
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1
    return md_state


def _force_err(f_a, f_b):
    d = np.linalg.norm(f_a - f_b, axis=1)
    scale = np.maximum(np.linalg.norm(f_b, axis=1), 1.0)
    rmsf = math.sqrt((f_b ** 2).sum(1).mean())
    return float((d / (1e-4 * scale + 1e-5 * rmsf)).max())


@pytest.mark.parametrize("inner_skin,temp", [(0.0, 300.0), (0.2, 600.0), (1.5, 900.0)])
def test_inner_list_forces_equal_plain_list_forces(mdx, inner_skin, temp):
    s = systems.water_box(16, seed=31, temp=temp)          # 12,288 atoms, hot: frequent re-pruning
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0, coulomb_mode=1, inner_skin=inner_skin)
    with mdx.MdState(s, cfg) as md:
        done = 0
        for burst in (1, 2, 3, 5, 7, 11, 13, 17, 19, 23):
            md.step(0.0005, None, burst)
            done += burst
            f_inner = md.forces().astype(np.float64)        # what the step loop computed (inner masks)
            md.energy()                                     # fresh evaluation over the plain list
            f_plain = md.forces().astype(np.float64)
            assert _force_err(f_inner, f_plain) < 1.0, f"after {done} steps"
        st = md.stats()
        assert st["prune_passes"] >= 10                     # at least the forced one of every burst
        assert 0 < st["n_inner_cluster_pairs"] < st["n_cluster_pairs"]


def test_long_burst_prunes_on_the_device(mdx):
    s = systems.water_box(16, seed=32, temp=500.0)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0, coulomb_mode=1, inner_skin=0.3)
    with mdx.MdState(s, cfg) as md:
        md.step(0.0005, None, 120)                          # ONE call: every pass after the first is device-triggered
        st = md.stats()
        assert st["prune_passes"] > 1 + st["rebuild_count"]
        f_inner = md.forces().astype(np.float64)
        md.energy()
        assert _force_err(f_inner, md.forces().astype(np.float64)) < 1.0


def test_trajectory_with_and_without_dual_list(mdx):
    s = systems.water_box(12, seed=33)
    L = np.asarray(s.box_hi, np.float64) - np.asarray(s.box_lo, np.float64)
    out = []
    for inner in (-1.0, 0.0):
        cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0, coulomb_mode=1, inner_skin=inner, nb_variant=5)
        with mdx.MdState(s, cfg) as md:
            md.step(0.0005, None, 60)
            out.append((md.positions().astype(np.float64), md.energy(), md.stats()))
    (x0, e0, st0), (x1, e1, st1) = out
    d = x1 - x0
    d -= np.round(d / L) * L
    assert math.sqrt((d ** 2).sum(1).mean()) < 2e-4
    assert abs(e1["potential"] - e0["potential"]) < 2e-5 * abs(e0["potential"]) + 0.05
    assert st0["prune_passes"] == 0 and st0["n_inner_cluster_pairs"] == 0
    assert st1["prune_passes"] > 0


@pytest.mark.parametrize("kind", ["rigid_tip3p", "opc"])
def test_constrained_and_virtual_site_runs(mdx, kind):
    """SHAKE corrections lengthen the path accumulators like the drift does; OPC's M site lies inside the triangle of
    its parents, so it never moves further than they do: the reference's default operating point (rigid 4-site water,
    dt = 2 fs, /root/reference src/prefs/mod.rs:203) walks the inner list too."""
    s = systems.water_box(10, seed=34, rigid=True, temp=400.0) if kind == "rigid_tip3p" else systems.opc_water_box(10, seed=35, temp=400.0)
    with mdx.MdState(s, MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.5, coulomb_mode=1, inner_skin=0.3)) as md:
        for burst in (1, 3, 7, 19, 30):
            md.step(0.002, None, burst)
            f_inner = md.forces().astype(np.float64)
            md.energy()
            assert _force_err(f_inner, md.forces().astype(np.float64)) < 1.0, f"{kind}: burst {burst}"
        st = md.stats()
        assert st["prune_passes"] > 5 + st["rebuild_count"]
        assert 0 < st["n_inner_cluster_pairs"] < st["n_cluster_pairs"]


def test_full_size_box(mdx):
    """BASELINE.json's 1,029,000-atom water box as generated (it heats towards 1300 K: the hardest case for a
    path-length bound): at every checkpoint all 1,029,000 step-loop forces equal a fresh plain-list evaluation."""
    s = systems.water1m()
    with mdx.MdState(s, MdConfig()) as md:                  # rc 10, skin 2, inner skin 0.5: the bench configuration
        for burst in (3, 20, 41, 64):
            md.step(0.0005, None, burst)
            f_inner = md.forces().astype(np.float64)
            md.energy()
            assert _force_err(f_inner, md.forces().astype(np.float64)) < 1.0, f"burst of {burst}"
        st = md.stats()
        assert st["prune_passes"] > 4 + st["rebuild_count"]
        assert st["n_inner_cluster_pairs"] < 0.85 * st["n_cluster_pairs"]


@pytest.mark.parametrize("geometric", [False, True])
def test_solvated_chain_with_exclusions_and_14_pairs(mdx, geometric):
    """A bonded chain in water: the masked run of the list (exclusions, 1-4 pairs, self pairs) stays in place in the
    inner list while the plain run is compacted; both combining rules."""
    s = systems.small_solvated(n_chain=400, box=44.0)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0, coulomb_mode=1, inner_skin=0.4,
                   combining_rule=1 if geometric else 0)
    with mdx.MdState(s, cfg) as md:
        md.minimize_energy(40)
        md.initialize_velocities(500.0, True, seed=3)
        for burst in (2, 9, 16, 33):
            md.step(0.0005, None, burst)
            f_inner = md.forces().astype(np.float64)
            md.energy()
            assert _force_err(f_inner, md.forces().astype(np.float64)) < 1.0, f"burst {burst}"
        st = md.stats()
        assert st["prune_passes"] >= 4 and st["n_masked_entries"] > 0
        assert 0 < st["n_inner_cluster_pairs"] < st["n_cluster_pairs"]


def test_default_buffer_tunes_itself_at_long_steps(mdx):
    """dt = 2 fs moves atoms four times as far per step: a fixed 0.5 A buffer would prune every other step and lose to
    the plain list.  The library-default buffer grows (or the handle returns to the plain list) until pruning passes are
    rare enough to pay; results stay those of the plain list throughout."""
    s = systems.water_box(12, seed=36, rigid=True, temp=330.0)
    with mdx.MdState(s, MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=2.0, coulomb_mode=1)) as md:   # inner_skin = 0: default
        md.step(0.002, None, 500)
        p0 = md.stats()["prune_passes"]
        for _ in range(4):
            md.step(0.002, None, 50)
            f_inner = md.forces().astype(np.float64)
            md.energy()
            assert _force_err(f_inner, md.forces().astype(np.float64)) < 1.0
        p1 = md.stats()["prune_passes"]
        assert (p1 - p0) / 200.0 < 0.5, "the dual list is still pruning on most steps"
