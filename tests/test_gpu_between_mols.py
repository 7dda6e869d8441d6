"""`SnapshotEnergyData.energy_potential_between_mols` (consumed at /root/reference src/properties/crystal.rs:347-370, 533; the
docking scorer's receptor - ligand term, src/docking/mod.rs:81-154) from the HIP path, through the C ABI
(mdx_set_energy_groups / mdx_energy_between_mols / mdx_snapshot_read_between_mols / mdx_single_point_between_mols), against the
fp64 oracle (orc_between_mols).

Tolerance per matrix element, as for the pair sums of tests/test_gpu_parity.py:  |dM| <= 2e-6 |M| + 1e-6 G + 1e-3, G = the oracle's
GROSS sum of |e_pair| over the same pairs (fp32 rounds the terms; an element is a sum with cancellation).  The matrix is stored
as f32 (the reference's type): + one f32 ulp of the element."""
import math
import threading

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1, "no GPU: the HIP path must run here, there is no fallback"
    return md_state


def groups_by_molecule(s):
    ms = np.asarray(s.mol_start, dtype=np.int64)
    g = np.zeros(s.n_atoms, np.uint8)
    for m in range(len(ms)):
        g[ms[m]:(ms[m + 1] if m + 1 < len(ms) else s.n_atoms)] = m
    return g, len(ms)


def three_groups(s):
    """receptor (molecule 0) / ligand (molecule 1) / solvent (the rest)"""
    ms = np.asarray(s.mol_start, dtype=np.int64)
    g = np.full(s.n_atoms, 2, np.uint8)
    g[:ms[1]] = 0
    g[ms[1]:ms[2]] = 1
    return g


def assert_matrix(m, mo, gross, what):
    m = np.asarray(m, np.float64)
    assert m.shape == mo.shape
    assert np.array_equal(m, m.T), f"{what}: the matrix must be symmetric"
    tol = 2e-6 * np.abs(mo) + 1e-6 * gross + 1e-3 + np.abs(mo) * 2.0 ** -23
    ratio = np.abs(m - mo) / tol
    k = np.unravel_index(np.argmax(ratio), ratio.shape)
    assert ratio[k] <= 1.0, f"{what}: element {k} gpu {m[k]!r} oracle {mo[k]!r}: {ratio[k]:.2f}x its tolerance {tol[k]:.2e}"


def test_receptor_ligand_solvent_matrix_matches_the_oracle(mdx, orc):
    s = systems.small_complex()
    for cfg in (MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5),
                MdConfig(lj_cutoff=9.0, coulomb_cutoff=8.0, skin=1.5, coulomb_mode=1),                       # reaction field, two cutoffs
                MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=2, ewald_alpha=0.35)):    # erfc real space (+ SPME)
        if cfg.coulomb_mode == 2:
            cfg.overrides = 0
        g = three_groups(s)
        with mdx.MdState(s, cfg) as md:
            assert md.set_energy_groups(g, 3) == 3
            m = md.energy_between_mols()
            e = md.energy()
            pos = md.positions()
            mo, gr = orc.between_mols(s, cfg, g, 3, pos=pos.astype(np.float64), use_cells=True)
            assert_matrix(m, mo, gr, f"small complex, coulomb mode {cfg.coulomb_mode}")
            # the matrix is made of the pair terms mdx_energy sums
            tot = float(np.asarray(m, np.float64)[np.triu_indices(3)].sum())
            want = e["lj"] + e["coulomb"] + e["lj14"] + e["coulomb14"]
            assert tot == pytest.approx(want, rel=2e-6, abs=2e-3)
            assert abs(m[0, 1]) > 1e-3 and abs(m[1, 2]) > 1e-3      # the ligand sees receptor and water
            # after a burst of steps (inner list in the step loop, the matrix pass walks the Verlet list)
            md.step(0.0005, None, 12)
            m2 = md.energy_between_mols()
            pos = md.positions()
            mo2, gr2 = orc.between_mols(s, cfg, g, 3, pos=pos.astype(np.float64), use_cells=True)
            assert_matrix(m2, mo2, gr2, f"small complex after 12 steps, coulomb mode {cfg.coulomb_mode}")


def test_full_list_kernels_give_the_same_matrix(mdx, orc):
    s = systems.small_complex()
    g = three_groups(s)
    ms = {}
    for variant in (5, 2, 1):
        cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, nb_variant=variant)
        with mdx.MdState(s, cfg) as md:
            md.set_energy_groups(g, 3)
            ms[variant] = md.energy_between_mols().astype(np.float64)
            pos = md.positions()
        mo, gr = orc.between_mols(s, cfg, g, 3, pos=pos.astype(np.float64), use_cells=True)
        assert_matrix(ms[variant], mo, gr, f"nb_variant {variant}")


def test_crystal_of_sixteen_molecules(mdx, orc):
    """One group per molecule (mol_start): the shape crystal.rs reads.  The strict upper triangle is the cohesive energy's
    numerator (crystal.rs:347-370): potential_nonbonded minus the intra-molecular parts on the diagonal."""
    s = systems.molecular_crystal()
    cfg = MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0)
    g, n = groups_by_molecule(s)
    assert n == 16
    with mdx.MdState(s, cfg) as md:
        assert md.set_energy_groups() == 16            # by molecule
        m = md.energy_between_mols()
        e = md.energy()
        pos = md.positions()
        mo, gr = orc.between_mols(s, cfg, g, n, pos=pos.astype(np.float64))
        assert_matrix(m, mo, gr, "crystal16")
        m64 = np.asarray(m, np.float64)
        upper = m64[np.triu_indices(n, 1)].sum()
        assert upper == pytest.approx(e["potential_nonbonded"] - np.trace(m64), rel=1e-5, abs=5e-3)
        assert upper == pytest.approx(mo[np.triu_indices(n, 1)].sum(), rel=1e-5, abs=5e-3)
        # snapshots carry the matrix of their step
        md.set_snapshot_cadence(5)
        md.step(0.0005, None, 10)
        snaps = md.snapshots
        assert len(snaps) == 2
        for sn in snaps:
            ms_ = sn["energy_data"]["energy_potential_between_mols"]
            mo_s, gr_s = orc.between_mols(s, cfg, g, n, pos=orc.wrap(s, sn["atom_posits"]).astype(np.float64))
            assert_matrix(ms_, mo_s, gr_s, f"crystal16 snapshot at step {sn['step']}")
        # an explicit map with fewer groups: rows / columns add up
        g4 = (g // 4).astype(np.uint8)
        assert md.set_energy_groups(g4, 4) == 4
        m4 = np.asarray(md.energy_between_mols(), np.float64)
        mfull = np.asarray(md.set_energy_groups() and md.energy_between_mols(), np.float64)
        for a in range(4):
            for b in range(4):
                blk = mfull[4 * a:4 * a + 4, 4 * b:4 * b + 4]
                want = blk.sum() if a != b else blk[np.triu_indices(4)].sum()
                assert m4[a, b] == pytest.approx(want, rel=1e-5, abs=2e-3)


def test_refusals(mdx):
    s = systems.molecular_crystal()
    with mdx.MdState(s, MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0)) as md:
        with pytest.raises(mdx.ParamError):
            md.energy_between_mols()                   # no groups set
        with pytest.raises(mdx.ParamError):
            md.set_energy_groups(np.full(s.n_atoms, 7, np.uint8), 4)      # group index out of range
        md.set_energy_groups()
        lib = mdx.load_library()
        import ctypes as C
        out = np.zeros(9, np.float32)
        assert lib.mdx_energy_between_mols(md._h, out.ctypes.data_as(C.POINTER(C.c_float)), 3) != 0      # n != number of groups
        # off again, explicitly, on a system WITH molecules ((NULL, 0) switches the by-molecule groups back on there)
        md.clear_energy_groups()
        assert lib.mdx_energy_group_count(md._h) == 0
        with pytest.raises(mdx.ParamError):
            md.energy_between_mols()
        assert lib.mdx_set_energy_groups(md._h, None, 5) != 0                   # a NULL map takes 0 or MDX_GROUPS_OFF
    # the stateless scorer with the by-molecule map: the caller's n_groups sizes matrix_out, so it must be the molecule count
    s.normalise()
    cs, cc = s.to_c(), MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0).to_c()
    from molchanica_amd._abi import CEnergies
    e = CEnergies()
    small = np.full(4, -7.0, np.float32)
    rc = lib.mdx_single_point_between_mols(C.byref(cs), C.byref(cc), 0, None, 2, C.byref(e), None, small.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc != 0 and (small == -7.0).all(), "a by-molecule request with n_groups != n_mols must be refused before anything is written"
    w = systems.water_box(8, seed=3)                   # 512 molecules: more than a byte can index
    with mdx.MdState(w, MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5)) as md:
        with pytest.raises(mdx.ParamError):
            md.set_energy_groups()


def test_c3_scorer_returns_the_receptor_ligand_energy(mdx, orc):
    """BASELINE config 3 through the stateless scorer: pose after pose, the ligand row of the receptor / ligand / solvent matrix
    against the oracle (rel 2e-6 + the gross term), the totals unchanged by asking for the matrix."""
    s = systems.complex50k()
    cfg = MdConfig()
    g = three_groups(s)
    lig = slice(int(s.mol_start[1]), int(s.mol_start[2]))
    rng = np.random.default_rng(23)
    base = s.pos.copy()
    mdx.release_single_point_cache()
    for pose in range(3):
        p = base.copy()
        if pose:
            c = p[lig].mean(0)
            ang = 0.05 * pose
            rot = np.array([[math.cos(ang), -math.sin(ang), 0], [math.sin(ang), math.cos(ang), 0], [0, 0, 1]])
            p[lig] = (p[lig] - c) @ rot.T + c + rng.normal(0, 0.15 * pose, 3) + rng.normal(0, 0.01, (50, 3))
        s.pos = p.astype(np.float32)
        e = mdx.compute_energy_snapshot(s, cfg, groups=g, n_groups=3)
        e_plain = mdx.compute_energy_snapshot(s, cfg)
        for k in ("potential", "lj", "coulomb", "bond"):
            assert e[k] == pytest.approx(e_plain[k], rel=1e-9, abs=1e-6)      # (two evaluations: the f32 atomics land in another order)
        pw = orc.wrap(s, s.pos)
        mo, gr = orc.between_mols(s, cfg, g, 3, pos=pw.astype(np.float64), use_cells=True)
        assert_matrix(e["energy_potential_between_mols"], mo, gr, f"complex50k pose {pose}")
        assert abs(mo[1, 0]) > 0.1 and abs(mo[1, 2]) > 0.1
    mdx.release_single_point_cache()


def test_matrix_on_four_virtual_ranks(mdx, orc):
    """Decomposed handle (2 x 2 x 1 ranks as threads over the in-process fabric): every rank returns the matrix of the whole
    box - a pair counts where it is evaluated, the raw sums are all-reduced."""
    from molchanica_amd.md_state import Fabric, MdState
    s = systems.small_complex(box=44.0)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1, chunk_steps=8)
    g = three_groups(s)
    with MdState(s, cfg) as md:
        md.set_energy_groups(g, 3)
        m1 = md.energy_between_mols().astype(np.float64)
        pos = md.positions()
    mo, gr = orc.between_mols(s, cfg, g, 3, pos=pos.astype(np.float64), use_cells=True)
    assert_matrix(m1, mo, gr, "one GPU")
    world = 4
    fabric = Fabric(world)
    res, errs = {}, []

    def run(rank):
        try:
            with MdState(s, cfg) as md:
                md.set_energy_groups(g, 3)
                md.comm_init_fabric(fabric, rank)
                m0 = md.energy_between_mols()
                md.step(0.0005, None, 10)
                res[rank] = (m0, md.energy_between_mols(), md.positions())
        except BaseException as e:   # pragma: no cover
            errs.append(e)
            fabric.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    for r in range(world):
        assert_matrix(res[r][0], mo, gr, f"rank {r} of 4, start")
        assert np.array_equal(res[r][0], res[0][0]) and np.array_equal(res[r][1], res[0][1])
    mo2, gr2 = orc.between_mols(s, cfg, g, 3, pos=orc.wrap(s, res[0][2]).astype(np.float64), use_cells=True)
    assert_matrix(res[0][1], mo2, gr2, "4 ranks after 10 steps")
