"""Pins the oracle's leapfrog and Langevin-middle integrators (`Integrator::{Leapfrog, LangevinMiddle{gamma}}`,
src/ui/panels/md.rs:296-305) - CPU only.

  I1  leapfrog started from v(-dt/2) = v(0) - dt/2 a(0) visits exactly the positions of velocity Verlet.
  I2  Langevin middle with gamma = 0 IS leapfrog.
  I3  with forces off, the friction + noise step drives every atom to the Maxwell distribution of the target
      temperature (exact for the discrete update: <v^2> -> kT/m), and is reproducible from (seed, step, atom).
  I4  the noise stream: unit variance, zero mean, no correlation between atoms or steps.
  I5  rigid water under Langevin middle keeps its constraints and its temperature.
"""
import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

KB = 0.0019872041
ACC = 418.4


def test_i1_leapfrog_positions_equal_velocity_verlet(orc):
    s = systems.water_box(4, seed=3)
    cfg = MdConfig(lj_cutoff=5.5, coulomb_cutoff=5.5, skin=0.5, coulomb_mode=1)
    dt, n = 0.0005, 25
    x_vv, v_vv, _ = orc.step(s, cfg, dt, n)
    f0, _ = orc.forces(s, cfg)
    v_half = s.vel.astype(np.float64) - 0.5 * dt * f0 * (ACC / s.mass.astype(np.float64))[:, None]
    x_lf, v_lf, _ = orc.step_integrator(s, cfg, dt, n, kind=1, vel=v_half)
    assert np.abs(x_lf - x_vv).max() < 1e-10
    # and the half-step velocity is the one velocity Verlet passes through: v(t) = v(t - dt/2) + dt/2 a(t)
    fn, _ = orc.forces(s, cfg, pos=x_lf)
    assert np.abs(v_lf + 0.5 * dt * fn * (ACC / s.mass.astype(np.float64))[:, None] - v_vv).max() < 1e-9


def test_i2_langevin_without_friction_is_leapfrog(orc):
    s = systems.water_box(4, seed=5)
    cfg = MdConfig(lj_cutoff=5.5, coulomb_cutoff=5.5, skin=0.5, coulomb_mode=1)
    a = orc.step_integrator(s, cfg, 0.0005, 15, kind=1)
    b = orc.step_integrator(s, cfg, 0.0005, 15, kind=2, gamma=0.0, temperature=300.0, seed=9)
    assert np.abs(a[0] - b[0]).max() < 1e-10 and np.abs(a[1] - b[1]).max() < 1e-9   # two half drifts vs one: rounding only


def test_i3_free_particles_thermalise_exactly(orc):
    s = systems.water_box(6, seed=1)
    s.vel = np.zeros_like(s.vel)
    cfg = MdConfig(lj_cutoff=5.0, coulomb_cutoff=5.0, skin=1.0, overrides=0x1 | 0x2 | 0x4 | 0x8)   # no forces at all
    x, v, e = orc.step_integrator(s, cfg, 0.002, 200, kind=2, gamma=20.0, temperature=350.0, seed=12)
    t = 2 * orc.kinetic(s, v) / (3 * s.n_atoms * KB)
    assert t == pytest.approx(350.0, rel=0.06)                      # 648 atoms: sigma_T / T = sqrt(2 / 3N) = 3 %
    m = s.mass.astype(np.float64)
    zo = v[m > 10] * np.sqrt(m[m > 10] / (KB * 350.0 * ACC))[:, None]   # standardised: unit normal per component
    assert abs(zo.mean()) < 0.1 and zo.std() == pytest.approx(1.0, rel=0.1)
    x2, v2, _ = orc.step_integrator(s, cfg, 0.002, 200, kind=2, gamma=20.0, temperature=350.0, seed=12)
    assert np.array_equal(v, v2)
    # split in two bursts with the step counter carried over: same trajectory
    xa, va, _ = orc.step_integrator(s, cfg, 0.002, 120, kind=2, gamma=20.0, temperature=350.0, seed=12)
    xb, vb, _ = orc.step_integrator(s, cfg, 0.002, 80, kind=2, gamma=20.0, temperature=350.0, seed=12, step0=120, pos=xa, vel=va)
    assert np.abs(vb - v).max() < 1e-12
    _, v3, _ = orc.step_integrator(s, cfg, 0.002, 200, kind=2, gamma=20.0, temperature=350.0, seed=13)
    assert not np.allclose(v3, v)


def test_i4_noise_stream_statistics(orc):
    g = np.array([[orc.langevin_normals(7, st, a) for a in range(400)] for st in range(40)])   # [step, atom, 3]
    flat = g.reshape(-1)
    assert abs(flat.mean()) < 0.02 and flat.std() == pytest.approx(1.0, rel=0.02)
    assert abs(np.mean(g[:-1] * g[1:])) < 0.02            # successive steps
    assert abs(np.mean(g[:, :-1] * g[:, 1:])) < 0.02      # neighbouring atoms
    assert abs(np.mean(g[..., 0] * g[..., 1])) < 0.02 and abs(np.mean(g[..., 0] * g[..., 2])) < 0.02
    assert np.abs(flat).max() < 6.0


def test_i5_rigid_water_langevin(orc):
    s = systems.water_box(4, seed=8, rigid=True)
    cfg = MdConfig(lj_cutoff=5.5, coulomb_cutoff=5.5, skin=0.5, coulomb_mode=1)
    x, v, e = orc.step_integrator(s, cfg, 0.002, 300, kind=2, gamma=25.0, temperature=300.0, seed=4)
    L = float(s.box_hi[0])
    d = x[0::3] - x[1::3]; d -= np.round(d / L) * L
    assert np.abs(np.linalg.norm(d, axis=1) - np.float32(systems.TIP3P["r_oh"])).max() < 1e-8
    t = 2 * orc.kinetic(s, v) / (orc.dof(s) * KB)
    assert 200.0 < t < 430.0    # 64 waters: sigma_T ~ 35 K, and the lattice start is still releasing heat
