"""Pins the CPU oracle (oracle/mdx_oracle.c) — CPU only.

The reference holds NO test, fixture or golden vector on the MD path (src/tests.rs:3-4 is an
empty `fn test_basic_forces() {}`) and its engine crate cannot be built here, so
the oracle is pinned by analytic known-answer tests (K1-K7) and self-consistency checks (V1-V5).
The pair arithmetic the reference's tree states in native code is pinned by running it (tests/test_reference_pin.py,
tests/test_gpu_reference_kernels.py); for everything the absent crate decides alone these tests are what stands in.
"""
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, MdSystem, systems

KE = 332.0637


def two_atoms(r, sigma=(3.0, 3.0), eps=(0.2, 0.2), q=(0.0, 0.0), **kw):
    return MdSystem(pos=[[0, 0, 0], [r, 0, 0]], mass=[12, 12], charge=q, lj_type=[0, 1],
                    lj_sigma=sigma, lj_eps=eps, **kw).normalise()


NOCUT = dict(lj_cutoff=0.0, coulomb_cutoff=0.0)


def test_k1_lj_minimum_and_zero(orc):
    # src/cuda/util.cu:92-115: E = 4 eps (s^12 - s^6), F = dir * 24 eps (2 s^12 - s^6)/r, dir = tgt - src
    sig, eps = 3.0, 0.2
    f, e = orc.forces(two_atoms(2 ** (1 / 6) * sig), MdConfig(**NOCUT))
    assert e["lj"] == pytest.approx(-eps, rel=1e-6)      # parameters travel as f32
    assert np.abs(f).max() < 1e-5
    f, e = orc.forces(two_atoms(sig), MdConfig(**NOCUT))
    assert e["lj"] == pytest.approx(0.0, abs=1e-6)
    # repulsive along tgt - src: atom 1 (at +x) is pushed to +x, atom 0 to -x
    assert f[1, 0] == pytest.approx(24 * eps / sig, rel=1e-6)
    assert f[0, 0] == pytest.approx(-24 * eps / sig, rel=1e-6)


def test_k1_lorentz_berthelot_and_geometric(orc):
    s = two_atoms(3.3, sigma=(3.0, 3.6), eps=(0.1, 0.4))
    _, e = orc.forces(s, MdConfig(**NOCUT))
    sg, ep = 3.3, 0.2
    assert e["lj"] == pytest.approx(4 * ep * ((sg / 3.3) ** 12 - (sg / 3.3) ** 6), rel=1e-6)
    _, e = orc.forces(s, MdConfig(combining_rule=1, **NOCUT))
    sg = math.sqrt(3.0 * 3.6)
    assert e["lj"] == pytest.approx(4 * ep * ((sg / 3.3) ** 12 - (sg / 3.3) ** 6), rel=2e-6)


def test_k2_coulomb_unit_charges(orc):
    # two charges +-1 e at 1 Å: E = -k_e, |F| = k_e, attractive (src/cuda/util.cu:53-63 with k_e explicit)
    s = two_atoms(1.0, eps=(0, 0), q=(1.0, -1.0))
    f, e = orc.forces(s, MdConfig(**NOCUT))
    assert e["coulomb"] == pytest.approx(-KE, rel=1e-7)
    assert f[0, 0] == pytest.approx(KE, rel=1e-7) and f[1, 0] == pytest.approx(-KE, rel=1e-7)
    # the reference toy kernel's softening 1e-6 Å^2 as a parameter
    f2, _ = orc.forces(s, MdConfig(softening_sq=1e-6, **NOCUT))
    assert f2[0, 0] == pytest.approx(KE / (1 + 1e-6), rel=1e-7)


def test_k2_shifted_cutoff_and_reaction_field(orc):
    box = dict(periodic=True, box_lo=(0, 0, 0), box_hi=(30, 30, 30))
    s = two_atoms(4.0, eps=(0, 0), q=(0.5, -0.4), **box)
    rc = 9.0
    f, e = orc.forces(s, MdConfig(lj_cutoff=rc, coulomb_cutoff=rc))
    assert e["coulomb"] == pytest.approx(KE * 0.5 * -0.4 * (1 / 4.0 - 1 / rc), rel=1e-6)
    assert f[0, 0] == pytest.approx(-KE * 0.5 * -0.4 / 16.0, rel=1e-6)
    f, e = orc.forces(s, MdConfig(lj_cutoff=rc, coulomb_cutoff=rc, coulomb_mode=1))
    krf = 1 / (2 * rc ** 3)
    assert e["coulomb"] == pytest.approx(KE * -0.2 * (1 / 4 + krf * 16 - 1.5 / rc), rel=1e-6)
    assert abs(f[0, 0]) == pytest.approx(KE * 0.2 * (1 / 16 - 2 * krf * 4), rel=1e-6)
    # beyond the cutoff: nothing
    s = two_atoms(9.5, q=(0.5, -0.4), **box)
    f, e = orc.forces(s, MdConfig(lj_cutoff=rc, coulomb_cutoff=rc))
    assert e["potential"] == 0.0 and np.abs(f).max() == 0.0


def test_k3_bond(orc):
    k, r0 = 300.0, 1.4
    s = two_atoms(r0 + 0.1, eps=(0, 0), bond_idx=[[0, 1]], bond_k=[k], bond_r0=[r0],
                  excl_offsets=[0, 1, 2], excl_idx=[1, 0])
    f, e = orc.forces(s, MdConfig(**NOCUT))
    assert e["bond"] == pytest.approx(0.01 * k, rel=1e-6)
    assert f[0, 0] == pytest.approx(0.2 * k, rel=1e-6)      # stretched: pulled together
    assert f[1, 0] == pytest.approx(-0.2 * k, rel=1e-6)


def test_k4_angle(orc):
    k, t0 = 80.0, 1.9
    th = t0 + 0.1
    pos = [[1.0, 0, 0], [0, 0, 0], [math.cos(th), math.sin(th), 0]]
    s = MdSystem(pos=pos, mass=[1, 1, 1], charge=[0, 0, 0], lj_type=[0, 0, 0], lj_sigma=[1.0], lj_eps=[0.0],
                 angle_idx=[[0, 1, 2]], angle_k=[k], angle_theta0=[t0]).normalise()
    f, e = orc.forces(s, MdConfig(**NOCUT))
    assert e["angle"] == pytest.approx(0.01 * k, rel=1e-5)
    # torque balance and |F_0| = dE/dtheta / r
    assert np.abs(f.sum(0)).max() < 1e-9
    assert np.linalg.norm(f[0]) == pytest.approx(2 * k * 0.1 / 1.0, rel=1e-5)


@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("phase", [0.0, math.pi])
@pytest.mark.parametrize("phi", [0.0, math.pi / 3, math.pi / 2, math.pi])
def test_k5_dihedral_closed_form(orc, n, phase, phi):
    # i-j-k-l with j-k along z; phi = 0 is cis
    pos = [[1.0, 0, 0], [0, 0, 0], [0, 0, 1.5], [math.cos(phi), math.sin(phi), 1.5]]
    v = 1.3
    s = MdSystem(pos=pos, mass=[1] * 4, charge=[0] * 4, lj_type=[0] * 4, lj_sigma=[1.0], lj_eps=[0.0],
                 dihedral_idx=[[0, 1, 2, 3]], dihedral_v=[v], dihedral_phase=[phase], dihedral_n=[n]).normalise()
    _, e = orc.forces(s, MdConfig(**NOCUT))
    assert e["dihedral"] == pytest.approx(v * (1 + math.cos(n * phi - phase)), abs=1e-6)


def test_k6_min_image(orc):
    # d = 0.6 L -> -0.4 L (src/cuda/util.cu:65-71); tie at exactly L/2 is measure-zero and left to rint
    L = 30.0
    box = dict(periodic=True, box_lo=(0, 0, 0), box_hi=(L, L, L))
    s = MdSystem(pos=[[1.0, 5, 5], [1.0 + 0.8 * L, 5, 5]], mass=[1, 1], charge=[1.0, -1.0], lj_type=[0, 0],
                 lj_sigma=[1.0], lj_eps=[0.0], **box).normalise()
    f, e = orc.forces(s, MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0))
    r = 0.2 * L
    assert e["coulomb"] == pytest.approx(-KE * (1 / r - 1 / 9.0), rel=1e-6)
    assert f[0, 0] == pytest.approx(-KE / r ** 2, rel=1e-6)   # attracted through the -x face


def test_k7_exclusions_and_14_scaling(orc):
    # 4-atom chain: 1-2 and 1-3 excluded, 1-4 scaled by 1/2 (LJ) and 1/1.2 (Coulomb)
    pos = np.array([[0, 0, 0], [1.5, 0, 0], [2.2, 1.3, 0], [3.7, 1.4, 0.4]], dtype=float)
    q = [0.3, -0.2, 0.1, -0.4]
    from molchanica_amd import topology as topo
    adj = topo.adjacency(4, [[0, 1], [1, 2], [2, 3]])
    off, idx, p14 = topo.exclusions_and_pairs14(4, adj)
    assert p14.tolist() == [[0, 3]]
    s = MdSystem(pos=pos, mass=[12] * 4, charge=q, lj_type=[0] * 4, lj_sigma=[3.2], lj_eps=[0.15],
                 excl_offsets=off, excl_idx=idx, pairs14_idx=p14).normalise()
    _, e = orc.forces(s, MdConfig(**NOCUT))
    r = np.linalg.norm(pos[0] - pos[3])
    sr6 = (3.2 / r) ** 6
    assert e["lj"] == 0.0 and e["coulomb"] == 0.0
    assert e["lj14"] == pytest.approx(0.5 * 4 * 0.15 * (sr6 * sr6 - sr6), rel=1e-5)
    assert e["coulomb14"] == pytest.approx(KE * q[0] * q[3] / r / 1.2, rel=1e-6)


def test_overrides_disable_terms(orc):
    s = systems.lig50()
    base = orc.forces(s, MdConfig(**NOCUT))[1]
    e = orc.forces(s, MdConfig(overrides=0x1, **NOCUT))[1]
    assert e["potential_bonded"] == 0.0 and e["lj14"] == 0.0 and e["lj"] == pytest.approx(base["lj"])
    e = orc.forces(s, MdConfig(overrides=0x2, **NOCUT))[1]
    assert e["coulomb"] == 0.0 and e["coulomb14"] == 0.0 and e["lj"] == pytest.approx(base["lj"])
    e = orc.forces(s, MdConfig(overrides=0x4, **NOCUT))[1]
    assert e["lj"] == 0.0 and e["lj14"] == 0.0 and e["coulomb"] == pytest.approx(base["coulomb"])


# ---- self-consistency ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["lig50", "solv_rf", "solv_ewald"])
def test_v1_force_is_minus_gradient(orc, case):
    if case == "lig50":
        s, cfg = systems.lig50(), MdConfig(**NOCUT)
    elif case == "solv_rf":   # reaction field: potential continuous at the cutoff
        s, cfg = systems.small_solvated(), MdConfig(lj_cutoff=9, coulomb_cutoff=9, coulomb_mode=1)
    else:
        s, cfg = systems.small_solvated(), MdConfig(lj_cutoff=9, coulomb_cutoff=9, coulomb_mode=2, ewald_alpha=0.35)
    x0 = s.pos.astype(np.float64)
    f, _ = orc.forces(s, cfg, pos=x0)
    rng = np.random.default_rng(0)
    h = 1e-5
    for i in rng.choice(s.n_atoms, 6, replace=False):
        for a in range(3):
            xp, xm = x0.copy(), x0.copy()
            xp[i, a] += h
            xm[i, a] -= h
            fd = -(orc.forces(s, cfg, pos=xp)[1]["potential"] - orc.forces(s, cfg, pos=xm)[1]["potential"]) / (2 * h)
            assert fd == pytest.approx(f[i, a], rel=2e-6, abs=2e-6)


def test_v2_net_force_and_torque_vanish_in_vacuum(orc):
    s = systems.lig50()
    f, _ = orc.forces(s, MdConfig(**NOCUT))
    assert np.abs(f.sum(0)).max() < 1e-9
    assert np.abs(np.cross(s.pos.astype(np.float64), f).sum(0)).max() < 1e-8


def test_v3_cell_list_equals_brute_force(orc):
    s = systems.small_solvated()
    for rl in (6.0, 10.5):
        a = orc.neighbor_list(s, rl, use_cells=False)
        b = orc.neighbor_list(s, rl, use_cells=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    cfg = MdConfig(lj_cutoff=9, coulomb_cutoff=9)
    f0, e0 = orc.forces(s, cfg, use_cells=False)
    f1, e1 = orc.forces(s, cfg, use_cells=True)
    assert np.abs(f0 - f1).max() < 1e-9 and e0["potential"] == pytest.approx(e1["potential"], abs=1e-8)
    # symmetry of the list
    off, idx = a
    pairs = set()
    for i in range(s.n_atoms):
        for j in idx[off[i]:off[i + 1]]:
            pairs.add((i, int(j)))
    assert all((j, i) in pairs for (i, j) in pairs)


def test_v4_nve_energy_conservation(orc):
    s = systems.water_box(5, seed=7, jitter=0.0)
    cfg = MdConfig(lj_cutoff=6.5, coulomb_cutoff=6.5, coulomb_mode=1, skin=1.0)  # RF: continuous potential
    x, v, e0 = orc.step(s, cfg, 0.00025, 0)
    e0["kinetic"] = orc.kinetic(s, v)
    x, v, e1 = orc.step(s, cfg, 0.00025, 200, pos=x, vel=v)
    drift = abs((e1["potential"] + e1["kinetic"]) - (e0["potential"] + e0["kinetic"]))
    # 0.05 ps; the LJ truncation at rc is the only non-smooth term left
    assert drift / s.n_atoms / 0.05 < 2e-2, drift


def test_v5_time_reversal(orc):
    s = systems.lig50()
    cfg = MdConfig(**NOCUT)
    x0, v0 = s.pos.astype(np.float64), s.vel.astype(np.float64)
    x1, v1, _ = orc.step(s, cfg, 2e-4, 50, pos=x0, vel=v0)
    x2, v2, _ = orc.step(s, cfg, -2e-4, 50, pos=x1, vel=v1)
    assert np.abs(x2 - x0).max() < 1e-9


def test_static_and_ext_forces(orc):
    s = systems.lig50()
    s.flags = np.zeros(50, np.uint8)
    s.flags[:5] = 1
    ext = np.zeros((50, 3))
    ext[10] = [5.0, 0, 0]
    x, v, _ = orc.step(s, MdConfig(**NOCUT), 2e-4, 5, ext=ext)
    assert np.array_equal(x[:5], s.pos[:5].astype(np.float64))
    f0 = orc.forces(s, MdConfig(**NOCUT))[0]
    f1 = orc.forces(s, MdConfig(**NOCUT), ext=ext)[0]
    assert np.allclose(f1 - f0, ext)


# ---- committed golden vectors --------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lig50", "water648"])
def test_oracle_reproduces_golden(orc, golden_dir, name):
    import os
    from tests.golden.make_golden import CASES
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    mk, kw = CASES[name]
    s, cfg = mk(), MdConfig(**kw)
    assert np.array_equal(s.pos, g["pos"])          # the seeded generator is part of the fixture
    f, e = orc.forces(s, cfg)
    assert np.abs(f - g["forces"]).max() < 1e-9
    for k in ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14"):
        assert e[k] == pytest.approx(float(g[f"e_{k}"]), rel=1e-12, abs=1e-10)
    if "nl_offsets" in g:
        off, idx = orc.neighbor_list(s, cfg.lj_cutoff + cfg.skin, use_cells=True)
        assert np.array_equal(off, g["nl_offsets"]) and np.array_equal(idx, g["nl_idx"])
