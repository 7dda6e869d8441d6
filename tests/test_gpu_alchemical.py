"""GPU parity of the alchemical window (`md.configure_alchemical_window(dev, mol_index, lambda)`,
src/properties/water_sol.rs:556) against the oracle pinned by tests/test_oracle_alchemical.py, and the TI
bookkeeping driven the way `run_hydration_ti_window` drives it (water_sol.rs:532-580)."""
import numpy as np
import pytest

from molchanica_amd import MdConfig, systems
from molchanica_amd import alchemical as A
from molchanica_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1
    return md_state


CFG = dict(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=1)


@pytest.mark.parametrize("lam", [0.0, 0.35, 1.0])
def test_forces_energies_and_dh_dlambda_match_the_oracle(mdx, orc, lam):
    from tests.test_gpu_parity import assert_forces, assert_energies
    s = systems.small_solvated()
    cfg = MdConfig(**CFG)
    lo, hi = int(s.mol_start[0]), int(s.mol_start[1])
    with mdx.MdState(s, cfg) as md:
        e_plain = md.energy()
        assert e_plain["dh_dlambda"] == 0.0 and e_plain["coupled_interaction"] == 0.0
        md.configure_alchemical_window(0, lam)
        pos = md.positions(); f = md.forces(); e = md.energy()
        md.configure_alchemical_window(0, -1.0)          # off again: the plain energies come back
        e_off = md.energy()
    try:
        orc.set_alchemical(lo, hi, lam)
        fo, eo = orc.forces(s, cfg, pos=pos.astype(np.float64), use_cells=True)
        slack = orc.cutoff_slack(s, cfg, pos=pos)
    finally:
        orc.set_alchemical(0, 0, -1.0)
    assert_forces(f, fo, slack, f"alchemical lambda={lam}")
    assert_energies(e, eo, s.n_atoms * 200, f"alchemical lambda={lam}")
    scale = abs(eo["lj"]) + abs(eo["coulomb"]) + abs(eo["cross"])
    assert e["dh_dlambda"] == pytest.approx(-eo["cross"], abs=2e-5 * scale + 0.02)
    assert e["coupled_interaction"] == pytest.approx((1 - lam) * eo["cross"], abs=2e-5 * scale + 0.02)
    assert e_off["potential"] == pytest.approx(e_plain["potential"], rel=1e-9, abs=1e-4) and e_off["dh_dlambda"] == 0.0


def test_ti_windows_end_to_end(mdx):
    """Equilibrate, clear, produce, collect - per window - then integrate: the shape of water_sol.rs:556-580.
    Physics check: the chain attracts water, so decoupling costs free energy; <dH/dlambda> falls in size as lambda -> 1."""
    s = systems.small_solvated()
    cfg = MdConfig(**CFG)
    windows = []
    for lam in (0.0, 0.5, 1.0):
        with mdx.MdState(s, cfg) as md:
            md.configure_alchemical_window(0, lam)
            md.set_thermostat(2, 300.0, 0.1, 10, seed=7)
            md.step(0.001, None, 100)                     # equilibration
            md.flush_snapshot_queues()
            md.set_snapshot_cadence(10)
            md.step(0.001, None, 200)                     # production
            snaps = md.snapshots
            assert len(snaps) == 20
            w = A.collect_window(lam, snaps)
            assert w.n_samples == 20 and np.isfinite(w.mean_dh_dl) and w.sem_dh_dl >= 0
            if lam == 0.0:
                u = A.mean_coupled_interaction_kcal(snaps)
                assert u == pytest.approx(-w.mean_dh_dl, rel=1e-9)      # (1 - 0) U_cross = -dH/dlambda
            if lam == 1.0:
                assert abs(A.mean_coupled_interaction_kcal(snaps)) < 1e-9
            windows.append(w)
    dg, sem = A.free_energy_ti_with_sem(windows)
    assert np.isfinite(dg) and sem >= 0
    assert windows[0].mean_dh_dl > 0, "removing an attractive solute-water interaction raises the energy"
    assert dg > 0


def test_alchemical_parameter_errors(mdx):
    s = systems.small_solvated()
    with mdx.MdState(s, MdConfig(**CFG)) as md:
        with pytest.raises(mdx.ParamError):
            md.configure_alchemical_window(10 ** 6, 0.5)
        with pytest.raises(mdx.ParamError):
            md.configure_alchemical_window(0, 1.5)
    pme = MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=0.4, overrides=0)
    with mdx.MdState(s, pme) as md:
        with pytest.raises(mdx.ParamError):
            md.configure_alchemical_window(0, 0.5)
    # the deterministic full-list variant is overridden while a window is active (the alchemical flavour exists for
    # the half-list kernel only) and comes back afterwards
    with mdx.MdState(s, MdConfig(nb_variant=2, **CFG)) as md:
        f0 = md.forces()
        md.configure_alchemical_window(0, 0.0)
        f1 = md.forces()
        md.configure_alchemical_window(0, -1.0)
        f2 = md.forces()
        assert np.abs(f1 - f0).max() < 1e-2 and np.array_equal(f2, f0)
