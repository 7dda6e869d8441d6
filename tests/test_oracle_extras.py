"""Pins the oracle's restatement of the SURVEY §8f rows (minimiser, velocity initialisation,
thermostats) with closed forms and statistics.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, MdSystem, systems

KB = 0.0019872041
NOCUT = dict(lj_cutoff=0.0, coulomb_cutoff=0.0)


def test_initialize_velocities_statistics_and_determinism(orc):
    s = systems.water_box(8, seed=1)
    v1 = orc.init_velocities(s, 300.0, True, 42)
    v2 = orc.init_velocities(s, 300.0, True, 42)
    v3 = orc.init_velocities(s, 300.0, True, 43)
    assert np.array_equal(v1, v2) and not np.array_equal(v1, v3)
    p = (v1 * s.mass[:, None].astype(np.float64)).sum(0)
    assert np.abs(p).max() < 1e-9                                  # zero COM drift
    t = 2 * orc.kinetic(s, v1) / (orc.dof(s) * KB)
    assert t == pytest.approx(300.0, rel=0.05)
    # per-species Maxwell-Boltzmann width
    vh = v1[1::3]
    assert vh.std() == pytest.approx(math.sqrt(KB * 300 * 418.4 / 1.008), rel=0.05)
    # static atoms stay at rest but do not shift the stream of the others
    s.flags = np.zeros(s.n_atoms, np.uint8)
    s.flags[10] = 1
    v4 = orc.init_velocities(s, 300.0, False, 42)
    v5 = orc.init_velocities(systems.water_box(8, seed=1), 300.0, False, 42)
    assert np.array_equal(v4[10], [0, 0, 0]) and np.array_equal(v4[11:], v5[11:])


def test_berendsen_lambda_closed_form(orc):
    lib = orc.lib()
    rng = C.c_uint64(0)
    nf, t0, tau, dt = 300.0, 300.0, 0.5, 0.01
    for t in (150.0, 300.0, 600.0):
        ke = 0.5 * nf * KB * t
        lam = lib.orc_thermostat_lambda(1, ke, nf, t0, tau, dt, C.byref(rng))
        assert lam == pytest.approx(math.sqrt(1 + dt / tau * (t0 / t - 1)), rel=1e-12)


def test_csvr_relaxes_to_target_with_canonical_fluctuations(orc):
    """<K'> = c K + (1-c) K0 for one application; the stationary distribution is the canonical
    (gamma) one: mean K0, relative variance 2/Nf."""
    lib = orc.lib()
    nf, t0 = 60.0, 300.0
    k0 = 0.5 * nf * KB * t0
    rng = C.c_uint64(7)
    c = math.exp(-0.1 / 0.5)
    ks = np.array([0.5 * k0 * lib.orc_thermostat_lambda(2, 0.5 * k0, nf, t0, 0.5, 0.1, C.byref(rng)) ** 2
                   for _ in range(20000)])
    assert ks.mean() == pytest.approx(c * 0.5 * k0 + (1 - c) * k0, rel=0.01)
    k, traj = k0, []
    for _ in range(40000):
        k *= lib.orc_thermostat_lambda(2, k, nf, t0, 0.05, 0.1, C.byref(rng)) ** 2
        traj.append(k)
    traj = np.array(traj[2000:])
    assert traj.mean() == pytest.approx(k0, rel=0.02)
    assert traj.var() / traj.mean() ** 2 == pytest.approx(2.0 / nf, rel=0.1)


def test_thermostatted_run_reaches_target(orc):
    s = systems.water_box(4, seed=2, jitter=0.0)
    cfg = MdConfig(lj_cutoff=5.0, coulomb_cutoff=5.0, skin=0.5, coulomb_mode=1)
    x, v, temps = orc.step_thermo(s, cfg, 0.0005, 400, 1, 250.0, 0.02, 10, 0, zero_com=True)
    assert abs(temps[-5:].mean() - 250.0) < 40.0
    p = (v * s.mass[:, None]).sum(0)
    assert np.abs(p).max() < 1e-6


def test_minimizer_harmonic_dimer_and_monotone_energy(orc):
    k, r0 = 300.0, 1.4
    s = MdSystem(pos=[[0, 0, 0], [1.9, 0, 0]], mass=[12, 12], charge=[0, 0], lj_type=[0, 0], lj_sigma=[1.0],
                 lj_eps=[0.0], bond_idx=[[0, 1]], bond_k=[k], bond_r0=[r0], excl_offsets=[0, 1, 2],
                 excl_idx=[1, 0]).normalise()
    x, e, it = orc.minimize(s, MdConfig(**NOCUT), 200, f_tol=1e-3)
    assert np.linalg.norm(x[0] - x[1]) == pytest.approx(r0, abs=1e-4) and e["bond"] < 1e-6 and it < 200
    s = systems.lig50()
    e0 = orc.forces(s, MdConfig(**NOCUT))[1]["potential"]
    last = e0
    for n in (5, 20, 60):
        _, e, it = orc.minimize(s, MdConfig(**NOCUT), n)
        assert e["potential"] <= last + 1e-12 and it == n
        last = e["potential"]
    assert last < 0.5 * e0
    # static atoms do not move
    s.flags = np.zeros(50, np.uint8)
    s.flags[:10] = 1
    x, _, _ = orc.minimize(s, MdConfig(**NOCUT), 20)
    assert np.array_equal(x[:10], s.pos[:10].astype(np.float64))


def test_minimizer_follows_an_external_pull_from_a_relaxed_structure(orc):
    """`md.minimize_energy(dev, iters, Some(forces))` (src/mol_alignment.rs:356): the accepted quantity is
    U - sum F_ext . x.  A harmonic dimer at rest pulled apart by +-F settles where k-bond force balances the pull:
    2 k (r - r0) = F  =>  r = r0 + F / (2 k); judged on the internal potential alone every move would be refused."""
    k, r0, F = 300.0, 1.4, 30.0
    s = MdSystem(pos=[[0, 0, 0], [r0, 0, 0]], mass=[12, 12], charge=[0, 0], lj_type=[0, 0], lj_sigma=[1.0],
                 lj_eps=[0.0], bond_idx=[[0, 1]], bond_k=[k], bond_r0=[r0], excl_offsets=[0, 1, 2],
                 excl_idx=[1, 0]).normalise()
    ext = np.array([[-F, 0, 0], [F, 0, 0]], np.float64)
    x, e, it = orc.minimize(s, MdConfig(**NOCUT), 300, f_tol=1e-3, ext=ext)
    assert np.linalg.norm(x[0] - x[1]) == pytest.approx(r0 + F / (2 * k), abs=1e-4)
    assert e["bond"] == pytest.approx(k * (F / (2 * k)) ** 2, rel=1e-3) and it < 300


# ---- constraints and virtual sites -------------------------------------------------
def test_shake_rattle_rigid_water_conserves_energy_at_2fs(orc):
    s = systems.water_box(4, seed=3, rigid=True)
    cfg = MdConfig(lj_cutoff=5.5, coulomb_cutoff=5.5, skin=0.5, coulomb_mode=1)
    x0 = s.pos.astype(np.float64)
    v0 = orc.constrain_velocities(s, x0, s.vel.astype(np.float64))
    b = s.constraint_idx.astype(int)
    d = x0[b[:, 0]] - x0[b[:, 1]]
    assert np.abs((d * (v0[b[:, 0]] - v0[b[:, 1]])).sum(1)).max() < 1e-9      # no velocity along a bond
    e0 = orc.forces(s, cfg, pos=x0)[1]["potential"] + orc.kinetic(s, v0)
    x, v, e = orc.step(s, cfg, 0.002, 150, pos=x0, vel=v0)
    L = s.box_hi[0]
    d = x[b[:, 0]] - x[b[:, 1]]
    d -= np.round(d / L) * L
    assert np.abs(np.linalg.norm(d, axis=1) - s.constraint_len).max() < 1e-9
    assert abs(e["potential"] + e["kinetic"] - e0) / s.n_atoms < 0.02        # 0.3 ps at 2 fs
    assert orc.dof(s) == 3 * s.n_atoms - s.constraint_idx.shape[0] - 3


def test_virtual_site_geometry_and_force_spreading(orc):
    s = systems.opc_water_box(3, seed=2)
    x = orc.vsite_construct(s, s.pos.astype(np.float64) + 0.0)
    o, h1, h2, m = x[0::4], x[1::4], x[2::4], x[3::4]
    assert np.allclose(np.linalg.norm(m - o, axis=1), 0.1594, atol=2e-4)        # |OM| of OPC
    bis = (h1 - o) + (h2 - o)
    assert np.allclose(np.cross(bis, m - o), 0.0, atol=1e-6)                    # on the bisector
    cfg = MdConfig(lj_cutoff=4.0, coulomb_cutoff=4.0, skin=0.5, coulomb_mode=1)
    f, e = orc.forces(s, cfg)
    assert np.abs(f[3::4]).max() == 0.0                                         # massless site keeps no force
    assert np.abs(f.sum(0)).max() < 1e-9                                        # spreading conserves the total
    # F = -dE/dx through the chain rule of the site construction
    x0 = s.pos.astype(np.float64)
    h = 1e-5
    for i in (0, 1, 6):
        for a in range(3):
            xp, xm = x0.copy(), x0.copy()
            xp[i, a] += h
            xm[i, a] -= h
            fd = -(orc.forces(s, cfg, pos=xp)[1]["potential"] - orc.forces(s, cfg, pos=xm)[1]["potential"]) / (2 * h)
            assert fd == pytest.approx(f[i, a], rel=2e-5, abs=2e-5)


def test_shake_projects_along_old_bonds_and_respects_static_atoms(orc):
    s = MdSystem(pos=[[0, 0, 0], [1.2, 0, 0]], mass=[12, 1], charge=[0, 0], lj_type=[0, 0], lj_sigma=[1.0], lj_eps=[0.0],
                 constraint_idx=[[0, 1]], constraint_len=[1.0], excl_offsets=[0, 1, 2], excl_idx=[1, 0]).normalise()
    x_old = s.pos.astype(np.float64)
    x, _, _ = orc.constrain_positions(s, x_old, x_old)
    assert np.linalg.norm(x[0] - x[1]) == pytest.approx(1.0, abs=1e-10)
    assert (x * s.mass[:, None]).sum(0) == pytest.approx((x_old * s.mass[:, None]).sum(0))   # COM fixed
    assert abs(x[0, 0]) == pytest.approx(0.2 / 13, rel=1e-6)                                   # mass weighting
    s.flags = np.array([1, 0], np.uint8)
    x, _, _ = orc.constrain_positions(s, x_old, x_old)
    assert np.array_equal(x[0], [0, 0, 0]) and x[1, 0] == pytest.approx(1.0, abs=1e-10)


def test_shrink_cell_rule(orc):
    """K: the cell rule of sol_shrinking_box.rs:765-774 - edges shrink by the amount, stop at the target's, centre kept;
    coordinates follow affinely (the centre is a fixed point, the faces map onto the new faces)."""
    lo, hi = np.array([0.0, 1.0, -2.0]), np.array([30.0, 21.0, 8.0])
    tlo, thi = np.array([5.0, 5.0, 0.0]), np.array([25.0, 24.9, 9.9])        # target edges 20, 19.9, 9.9
    pos = np.array([[15.0, 11.0, 3.0], [0.0, 1.0, -2.0], [30.0, 21.0, 8.0]])
    nlo, nhi, x, shrank = orc.shrink_cell_towards(lo, hi, tlo, thi, 0.5, pos)
    assert shrank
    assert np.allclose(nhi - nlo, [29.5, 19.9, 9.9], atol=1e-5)              # y stops at the target edge, z likewise
    assert np.allclose(0.5 * (nlo + nhi), 0.5 * (lo + hi), atol=1e-5)
    assert np.allclose(x[0], pos[0], atol=1e-5) and np.allclose(x[1], nlo, atol=1e-4) and np.allclose(x[2], nhi, atol=1e-4)
    nlo2, nhi2, _, shrank2 = orc.shrink_cell_towards(tlo, thi, tlo, thi, 0.5, pos)
    assert not shrank2 and np.allclose(nlo2, tlo) and np.allclose(nhi2, thi)


def test_oracle_handles_rigid_waters_wrapped_across_box_faces(orc):
    """The GPU tests that found the virtual-site and halo image bugs of round 2 lean on the oracle for atom-wise wrapped
    rigid waters: it takes the minimum image per atom pair and per constraint / site vector, so a water whose sites sit on
    both sides of a face is the same molecule to it.  Energies and forces are invariant under wrapping, and 150 steps of
    NVE at dt 2 fs conserve the total energy from the wrapped state."""
    import numpy as np
    from molchanica_amd import MdConfig, systems
    s = systems.opc_water_box(5, seed=21)                         # 125 OPC waters, 15.5 A box
    L = np.array(s.box_hi, dtype=np.float64)
    cfg = MdConfig(lj_cutoff=6.0, coulomb_cutoff=6.0, skin=1.0, coulomb_mode=1)
    x_rel, v, _ = orc.step_thermo(s, cfg, 0.001, 400, 1, 300.0, 0.02, 1, 0)      # off the random-orientation lattice first
    x_whole = x_rel + 1.3
    for k in range(0, x_whole.shape[0], 4):                                       # every water whole, next to its oxygen ...
        d = x_whole[k + 1:k + 4] - x_whole[k]
        x_whole[k + 1:k + 4] -= np.round(d / L) * L
    x_wrapped = np.mod(x_whole, L)                                                # ... and wrapped atom by atom
    w = x_wrapped.reshape(-1, 4, 3)
    assert (np.abs(w[:, 1:] - w[:, :1]).max(axis=(1, 2)) > 0.5 * L[0]).sum() >= 10
    f0, e0 = orc.forces(s, cfg, pos=x_whole)
    f1, e1 = orc.forces(s, cfg, pos=x_wrapped)
    # (the oracle decides cut-off membership on canonical fp32 distances: a pair within rounding of the cut-off may flip)
    assert np.abs(f0 - f1).max() < 1e-3 and abs(e0["potential"] - e1["potential"]) < 1e-2
    t0 = e1["potential"] + orc.kinetic(s, v)
    x, v, e = orc.step(s, cfg, 0.002, 150, pos=x_wrapped, vel=v)
    assert abs(e["potential"] + orc.kinetic(s, v) - t0) < 0.02 * orc.kinetic(s, v)
