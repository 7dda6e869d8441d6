"""GPU parity of the SPME reciprocal sum: against the numpy SPME restatement
on the same mesh (tight) and against the textbook Ewald sum (loose: mesh discretisation)."""
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems
from molchanica_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1
    return md_state


def excluded_pairs(s):
    ii = np.repeat(np.arange(s.n_atoms), np.diff(s.excl_offsets.astype(np.int64)))
    jj = s.excl_idx.astype(np.int64)
    m = ii < jj
    pairs = np.stack([ii[m], jj[m]], 1)
    if s.pairs14_idx.shape[0]:
        pairs = np.concatenate([pairs, s.pairs14_idx.astype(np.int64)])
    return pairs


@pytest.mark.parametrize("which,side_stream", [("water", "0"), ("chain", "0"), ("chain", "1")])
def test_spme_matches_numpy_restatement_and_ewald(mdx, orc, which, side_stream, monkeypatch):
    """side_stream: the reciprocal-space chain beside the pair kernel on its own stream (the default from 65 k atoms up)
    or on the handle's stream (the default below) - MDX_PME_OVERLAP is read when a handle sets its mesh up."""
    from oracle import pme_ref as P
    monkeypatch.setenv("MDX_PME_OVERLAP", side_stream)
    s = systems.water_box(6, seed=3) if which == "water" else systems.small_solvated()
    L = float(s.box_hi[0])
    beta, grid = 0.40, (24, 24, 24) if which == "water" else (32, 32, 32)
    base = dict(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=beta)
    cfg_real = MdConfig(overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED, **base)
    cfg_full = MdConfig(overrides=0, pme_grid=grid, **base)
    with mdx.MdState(s, cfg_real) as md:
        pos = md.positions()
        f_real = md.forces().astype(np.float64)
        e_real = md.energy()
    with mdx.MdState(s, cfg_full) as md:
        f_full = md.forces().astype(np.float64)
        e_full = md.energy()
        assert np.array_equal(md.positions(), pos)
        md.step(0.0005, None, 20)                       # steps with the mesh in the loop
        e20 = md.energy()
    assert e_full["coulomb"] == pytest.approx(e_real["coulomb"], rel=1e-6, abs=1e-3) and e_real["coulomb_recip"] == 0.0
    box = np.full(3, L)
    q = s.charge.astype(np.float64)
    e_ref, f_ref = P.spme_recip(pos.astype(np.float64), q, (0, 0, 0), box, beta, grid, 4)
    e_x, f_x = P.excluded_pair_correction(pos.astype(np.float64), q, excluded_pairs(s), box, beta)
    e_ref += e_x + P.ewald_self_energy(q, beta) + P.ewald_background_energy(q, box, beta)
    f_ref += f_x
    f_rec = f_full - f_real
    err = math.sqrt(((f_rec - f_ref) ** 2).sum(1).mean()) / math.sqrt((f_ref ** 2).sum(1).mean())
    assert err < 2e-4, f"reciprocal force rms error {err:.2e} vs the numpy SPME on the same mesh"
    assert e_full["coulomb_recip"] == pytest.approx(e_ref, rel=2e-5, abs=5e-2)
    assert e_full["potential"] == pytest.approx(e_real["potential"] + e_full["coulomb_recip"], rel=1e-7, abs=1e-3)
    # textbook sum: only the mesh error separates the two
    e_d, f_d = P.ewald_recip_direct(pos.astype(np.float64), q, box, beta)
    f_d += f_x
    err_d = math.sqrt(((f_rec - f_d) ** 2).sum(1).mean()) / math.sqrt((f_d ** 2).sum(1).mean())
    assert err_d < 2e-2, err_d
    assert np.abs(f_full.sum(0)).max() < 0.5                                   # momentum (mesh: not exact)
    # the jittered lattice releases ~2 kcal/mol/atom in these 20 steps; same bound as the cutoff runs
    assert abs((e20["potential"] + e20["kinetic"]) - (e_full["potential"] + e_full["kinetic"])) / s.n_atoms < 0.05


@pytest.mark.parametrize("grid,edge,cap,side", [((24, 24, 24), None, None, False), ((50, 36, 30), "16", None, False), ((50, 36, 30), "11", None, True),
                                                ((27, 20, 45), "8", None, False), ((32, 32, 32), None, "8", False), ((16, 8, 12), "16", "8", True)])
def test_brick_spread_equals_the_tile_spread(mdx, grid, edge, cap, side, monkeypatch):
    """The charge spread of a single-GPU handle (mdx_pme.hip "Brick spread": bin -> canvas -> combine, no global atomics) and the
    gather through the same bricks (pme_gather_brick_kernel) against the tile kernel and the per-slot gather they replaced, on meshes
    whose edges the bricks do not divide, with every brick edge, with buckets so small that most atoms travel through the overflow
    lists, and (side) with the chain on its side stream, where the reciprocal force has an array of its own."""
    s = systems.small_solvated()
    base = dict(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=0.4, overrides=0, pme_grid=grid)
    out = {}
    for arm in ("tile", "brick"):
        monkeypatch.setenv("MDX_PME_SPREAD_BRICK", "0" if arm == "tile" else "1")
        monkeypatch.setenv("MDX_PME_OVERLAP", "1" if (side and arm == "brick") else "0")
        for k, v in (("MDX_PME_BRICK_EDGE", edge), ("MDX_PME_BRICK_CAP", cap)):
            monkeypatch.delenv(k, raising=False)
            if v is not None and arm == "brick":
                monkeypatch.setenv(k, v)
        with mdx.MdState(s, MdConfig(**base)) as md:
            f = md.forces().astype(np.float64)
            e = md.energy()
            md.step(0.0005, None, 12)
            out[arm] = (f, e, md.positions().astype(np.float64), md.pme_brick_overflows())
    (ft, et, pt, ot), (fb, eb, pb, ob) = out["tile"], out["brick"]
    assert ot == 0 and (ob > 0) == (cap is not None), (ot, ob)
    assert eb["coulomb_recip"] == pytest.approx(et["coulomb_recip"], rel=5e-6)      # (fp32 mesh, sums in another order)
    scale = np.maximum(np.abs(ft).max(1), 1.0)
    assert (np.abs(fb - ft).max(1) / scale).max() < 2e-5          # fp32 sums in another order on the mesh
    assert np.abs(pb - pt).max() < 2e-4


@pytest.mark.parametrize("grid", [(27, 20, 45), (50, 36, 30), (48, 24, 40), (30, 30, 30), (64, 64, 64), (20, 96, 10), (480, 16, 20)])      # (480: more than 64 KB of LDS per workgroup)
def test_fused_x_pass_equals_the_library_transform(mdx, grid, monkeypatch):
    """pme_xpass_solve_kernel (batched 2-D hipFFT + hand-written x pass with the solve inside; radices 4, 2, 3, 5, padded rows) against
    hipFFT's 3-D plan + pme_solve_kernel on the same handle inputs: energies (incl. the virial's pressure) and every force."""
    s = systems.small_solvated()
    base = dict(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=0.4, overrides=0, pme_grid=grid)
    out = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("MDX_PME_XPASS", arm)
        with mdx.MdState(s, MdConfig(**base)) as md:
            out[arm] = (md.forces().astype(np.float64), md.energy())
    (f0, e0), (f1, e1) = out["0"], out["1"]
    assert e1["coulomb_recip"] == pytest.approx(e0["coulomb_recip"], rel=2e-6)
    assert e1["pressure"] == pytest.approx(e0["pressure"], rel=1e-5, abs=1e-3)
    assert np.abs(f1 - f0).max() <= 2e-5 * max(1.0, np.abs(f0).max())


@pytest.mark.parametrize("beta,rc", [(0.25, 9.0), (0.30, 10.0), (0.34, 9.0), (0.42, 8.0), (0.50, 7.5), (0.30, 12.0)])
def test_ewald_force_table_equals_the_closed_form(mdx, beta, rc, monkeypatch):
    """The force-only Ewald flavour of the pair kernel (CM_EWALD_TAB: g(r^2) of qq (1/r^3 - g) from the bit-indexed LDS table of
    parabolas, mdx_pair_dev.h) against the closed form it replaces (erfc by A&S 7.1.26, MDX_EWALD_TABLE=0) on the same handle
    inputs, real space only, over the betas and cut-offs a caller may configure; and through a short trajectory."""
    s = systems.small_solvated(seed=17)
    cfg = MdConfig(lj_cutoff=rc, coulomb_cutoff=rc, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=beta,
                   overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED)
    out = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("MDX_EWALD_TABLE", arm)
        with mdx.MdState(s, cfg) as md:
            f = md.forces().astype(np.float64)
            md.step(0.0005, None, 20)
            out[arm] = (f, md.positions().astype(np.float64))
    (f0, p0), (f1, p1) = out["0"], out["1"]
    # per pair the table is good to 2e-6 of the pair's force and the closed form to 1.5e-7 in erfc; an atom sums ~200-400 pairs
    err = np.abs(f1 - f0).max(1) / np.maximum(np.abs(f0).max(1), 1.0)
    assert err.max() < 2e-5, err.max()
    assert np.abs(p1 - p0).max() < 1e-4


def test_spme_follows_the_box_and_rejects_bad_setups(mdx):
    s = systems.water_box(6, seed=3)
    cfg = MdConfig(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=0.4,
                   overrides=0)
    with mdx.MdState(s, cfg) as md:
        e0 = md.energy()
        L = np.array(s.box_hi)
        md.set_positions(md.positions() * 1.01)
        md.set_cell((0, 0, 0), tuple(L * 1.01))
        e1 = md.energy()
        assert e1["volume"] == pytest.approx(e0["volume"] * 1.01 ** 3, rel=1e-5)
        assert np.isfinite(e1["coulomb_recip"]) and e1["coulomb_recip"] != e0["coulomb_recip"]
    with pytest.raises(mdx.ParamError):
        mdx.MdState(s, MdConfig(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD,
                                ewald_alpha=0.4, overrides=0, pme_order=6))


@pytest.mark.parametrize("brick", ["0", "1"])
def test_large_like_signed_charges_on_a_coarse_mesh(mdx, brick, monkeypatch):
    """The LDS canvases of the charge spread accumulate in 32-bit fixed point (mdx_pme.hip, PME_FIX).  The scale is chosen per handle
    from the largest |q sqrt(k_e)| of the system: charges of +-4 e on a 2.3 A mesh - more spline-weighted charge per mesh point than
    the fixed 2^25 of round 4 could hold (it wrapped, silently, beyond +-3.5 e) - must give the mesh the fp64 restatement gives."""
    from oracle import pme_ref as P
    monkeypatch.setenv("MDX_PME_SPREAD_BRICK", brick)
    s = systems.water_box(6, seed=4)
    rng = np.random.default_rng(2)
    q = s.charge.copy()
    big = rng.choice(s.n_atoms, size=40, replace=False)
    q[big] = 4.0
    q[rng.choice(np.setdiff1d(np.arange(s.n_atoms), big), size=40, replace=False)] = -4.0
    q -= q.mean()                                   # neutral cell
    s.charge = q.astype(np.float32)
    L = float(s.box_hi[0])
    beta, grid = 0.35, (8, 8, 8)                    # 18.6 A / 8 = 2.3 A per mesh point (the smallest mesh the library takes): many atoms per point
    base = dict(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=beta)
    with mdx.MdState(s, MdConfig(overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED, **base)) as md:
        pos = md.positions()
        f_real = md.forces().astype(np.float64)
    with mdx.MdState(s, MdConfig(overrides=0, pme_grid=grid, **base)) as md:
        f_full = md.forces().astype(np.float64)
        e_full = md.energy()
    box = np.full(3, L)
    q64 = s.charge.astype(np.float64)
    e_ref, f_ref = P.spme_recip(pos.astype(np.float64), q64, (0, 0, 0), box, beta, grid, 4)
    e_x, f_x = P.excluded_pair_correction(pos.astype(np.float64), q64, excluded_pairs(s), box, beta)
    e_ref += e_x + P.ewald_self_energy(q64, beta) + P.ewald_background_energy(q64, box, beta)
    f_ref += f_x
    f_rec = f_full - f_real
    err = math.sqrt(((f_rec - f_ref) ** 2).sum(1).mean()) / math.sqrt((f_ref ** 2).sum(1).mean())
    assert err < 3e-4, f"reciprocal force rms error {err:.2e} vs the numpy SPME on the same mesh"
    assert e_full["coulomb_recip"] == pytest.approx(e_ref, rel=3e-5, abs=5e-2)
