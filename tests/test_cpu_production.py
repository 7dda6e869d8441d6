"""Pins the fp32 "production mode" CPU baseline (oracle/cpu_production.c: half Verlet list reused across steps,
OpenMP; what bench.py reports as cpu_baseline kind "port-production") against the fp64 oracle."""
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems


@pytest.fixture(scope="module")
def prod():
    from oracle import cpu_production
    cpu_production.lib()
    return cpu_production


@pytest.mark.parametrize("mode", [0, 1])
def test_forces_and_energies_match_the_fp64_oracle(prod, orc, mode):
    s = systems.small_solvated()
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=mode)
    f, e = prod.forces(s, cfg)
    fo, eo = orc.forces(s, cfg, pos=s.pos.astype(np.float64))
    slack = orc.cutoff_slack(s, cfg, pos=s.pos)
    err = np.linalg.norm(f - fo, axis=1)
    tol = 2e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + 1e-4 * math.sqrt((fo ** 2).sum(1).mean()) + slack
    assert (err <= tol).all(), float((err / tol).max())
    for k in ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14"):
        assert e[k] == pytest.approx(eo[k], rel=2e-5, abs=2e-2), k


def test_split_cutoffs_and_water_box(prod, orc):
    s = systems.water_box(8, seed=2)
    cfg = MdConfig(lj_cutoff=8.0, coulomb_cutoff=9.5, skin=1.0)
    f, e = prod.forces(s, cfg)
    fo, eo = orc.forces(s, cfg, pos=s.pos.astype(np.float64))
    err = np.linalg.norm(f - fo, axis=1)
    tol = 2e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + 1e-4 * math.sqrt((fo ** 2).sum(1).mean()) + orc.cutoff_slack(s, cfg, pos=s.pos)
    assert (err <= tol).all()
    assert e["lj"] == pytest.approx(eo["lj"], rel=2e-5, abs=2e-2) and e["coulomb"] == pytest.approx(eo["coulomb"], rel=2e-5, abs=2e-2)


def test_trajectory_with_list_reuse_follows_the_oracle(prod, orc):
    """60 steps of a hot box: the list is reused until an atom has moved skin/2, rebuilt a few times on the way; the
    fp32 trajectory stays on the fp64 oracle's (which searches afresh at every step)."""
    s = systems.small_solvated(n_chain=240, box=30.0)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.0, coulomb_mode=1)
    x, v, e, rebuilds = prod.run(s, cfg, 0.0005, 60, energy_every=60)
    xo, vo, eo = orc.step(s, cfg, 0.0005, 60, pos=s.pos.astype(np.float64), vel=s.vel.astype(np.float64), use_cells=True)
    L = np.array(s.box_hi) - np.array(s.box_lo)
    d = x - xo
    d -= np.round(d / L) * L
    assert math.sqrt((d ** 2).sum(1).mean()) < 2e-3
    assert 2 <= rebuilds < 30, rebuilds
    assert e["kinetic"] == pytest.approx(eo["kinetic"], rel=1e-3)
    pot = lambda q: sum(q[k] for k in ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14"))
    assert pot(e) == pytest.approx(pot(eo), rel=1e-4, abs=0.5)


def test_refuses_what_it_does_not_implement(prod):
    with pytest.raises(ValueError):
        prod.forces(systems.lig50(), MdConfig(lj_cutoff=0.0, coulomb_cutoff=0.0))
    with pytest.raises(ValueError):
        prod.forces(systems.small_solvated(), MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=2, ewald_alpha=0.3))


def test_rigid_and_virtual_site_systems_are_refused():
    """The baseline restates neither constraints nor virtual sites: it must refuse such a system instead of timing a different one."""
    import pytest
    from molchanica_amd import MdConfig, systems
    from oracle import cpu_production as cp
    cfg = MdConfig(lj_cutoff=6.0, coulomb_cutoff=6.0, skin=1.0)
    for s in (systems.water_box(6, seed=3, rigid=True), systems.opc_water_box(6, seed=3)):
        with pytest.raises(ValueError):
            cp.forces(s, cfg)
        with pytest.raises(ValueError):
            cp.run(s, cfg, 0.0005, 2)
