"""The oracle against RECORDED outputs of the reference's own pair kernels (tests/golden/ref_pair_kernels.npz: written on the
MI355X by tests/golden/make_ref_pair_kernels.py from /root/reference/src/cuda/cuda.cu + util.cu compiled as they lie).
CPU-only: this pin holds on a checkout that has neither /root/reference nor oracle/_ref.  The live kernels are held against
the same fixture - and the engine against both - in tests/test_gpu_reference_kernels.py."""
import numpy as np
import pytest

from molchanica_amd import MdConfig, MdSystem

from . import ref_cases as rc


@pytest.fixture(scope="module")
def fx():
    f = rc.load_fixture()
    assert f is not None, "tests/golden/ref_pair_kernels.npz is part of the repository"
    return f


def _box(ext):
    return MdSystem(pos=[[0.0, 0.0, 0.0]], mass=[12.0], charge=[0.0], lj_type=[0], lj_sigma=[3.0], lj_eps=[0.0], periodic=True,
                    box_lo=(0, 0, 0), box_hi=tuple(float(x) for x in ext)).normalise()


def test_canonical_minimum_image_equals_the_references_on_every_tie(orc, fx):
    """`min_image` (util.cu:65-71) vs the arithmetic inside the oracle's r2_canonical (mdx_oracle.c:73), value by value:
    d = +-L/2, +-3L/2, +-5L/2 (rintf: ties to even), one ulp either side, on each axis and on all three.
    The IMAGE (n = rintf(d / L)) must be the same in every case.  The VALUE is the same up to one rounding: the reference's
    file is compiled with the compilers' default contraction (nvcc -fmad, hipcc -ffp-contract=fast), which evaluates d - n L
    as ONE fma, where the oracle and the engine's list kernels spell two roundings (-ffp-contract=off: bit-exact neighbour
    lists need a contraction-proof form).  So: reference == fl(d - n L) with the oracle's n, bit for bit; oracle within
    half an ulp of the product n L of it (identical wherever n L is exact in fp32: n = 0, powers of two)."""
    cases = rc.min_image_cases()
    want = fx["min_image"]
    assert len(cases) == len(want)
    boxes = {}
    ties = differ = 0
    for (ext, d), w in zip(cases, want):
        s = boxes.setdefault(tuple(ext), _box(ext))
        got = orc.min_image_f32(s, d)
        n = np.rint((d.astype(np.float64) - got.astype(np.float64)) / ext.astype(np.float64))      # the image the oracle took
        assert np.array_equal(n, np.rint(d / ext).astype(np.float64)), (ext, d, n)
        fma_form = (d.astype(np.float64) - n * ext.astype(np.float64)).astype(np.float32)          # exact product, one rounding
        assert np.array_equal(fma_form.view(np.uint32), w.view(np.uint32)), ("the reference took another image", ext, d, w, fma_form)
        # the rounding the contraction saves is that of the product n L: half a unit in ITS last place (+ the final one)
        bound = 0.5 * np.spacing(np.abs(n * ext.astype(np.float64)).astype(np.float32)).astype(np.float64) + np.spacing(np.abs(w)).astype(np.float64)
        assert (np.abs(got.astype(np.float64) - w.astype(np.float64)) <= bound).all(), (ext, d, got, w)
        differ += int(not np.array_equal(got, w))
        r2 = orc.r2_canonical(s, d, np.zeros(3, np.float32))
        assert r2 == pytest.approx(float((got.astype(np.float64) ** 2).sum()), rel=3e-7)
        ties += int(np.any(np.abs(np.abs(d / ext) % 1.0 - 0.5) == 0.0))
    assert ties >= 60          # the sweep really holds exact ties
    assert differ <= len(cases) // 10
    # a tie stays on the even side: d = L/2 -> L/2 (rint(0.5) = 0), d = 3L/2 -> -L/2 (rint(1.5) = 2)
    s = _box([20.0, 30.0, 40.0])
    assert np.array_equal(orc.min_image_f32(s, [10.0, 45.0, -20.0]), np.array([10.0, -15.0, -20.0], np.float32))


def test_production_path_sums_of_dhfr23k_equal_the_references_arithmetic(orc, fx):
    """Everything the production path adds to the bare formulas - the cutoff filter, the periodic image, exclusions and 1-4
    removal, Lorentz-Berthelot tables, k_e - enters through the SOURCE SET: for 300 targets of dhfr23k the oracle's
    nonbonded force must equal lj_force_kernel + k_e * coulomb_force_kernel of the reference on that target's pre-imaged,
    cutoff-filtered sources (k_e = 332.0637 factored out: the reference's kernel has no unit constant, util.cu:53-63)."""
    s, cfg, pos, targets, cases = rc.dhfr_case()
    assert np.array_equal(targets, fx["dhfr_targets"]) and np.array_equal([len(c["nb"]) for c in cases], fx["dhfr_n_src"])
    assert fx["dhfr_n_src"].min() > 150 and fx["dhfr_n_src"].mean() > 300      # ~180-480 sources inside 10 A (the chain is less dense than water)
    f_ref = fx["dhfr_f_lj"].astype(np.float64) + rc.KE * fx["dhfr_f_coul_k1"].astype(np.float64)
    f_orc, _ = orc.forces(s, cfg, pos=pos.astype(np.float64), use_cells=True)
    f_t = np.asarray(f_orc)[targets]
    # scale: the terms of the sum, not the (partly cancelling) sum - fp32 reference arithmetic over ~400 sources
    scale = np.maximum(np.linalg.norm(f_ref, axis=1), 5.0)
    err = np.linalg.norm(f_t - f_ref, axis=1)
    assert (err <= 5e-5 * scale).all(), float((err / scale).max())
    assert np.linalg.norm(f_ref, axis=1).max() > 20.0
    # the two parts separately (LJ only / Coulomb only through the overrides)
    f_lj, _ = orc.forces(s, MdConfig(lj_cutoff=10.0, coulomb_cutoff=10.0, skin=2.0, overrides=cfg.overrides | 0x2), pos=pos.astype(np.float64), use_cells=True)
    e_lj = np.linalg.norm(np.asarray(f_lj)[targets] - fx["dhfr_f_lj"], axis=1)
    assert (e_lj <= 5e-5 * np.maximum(np.linalg.norm(fx["dhfr_f_lj"], axis=1), 1.0)).all(), float(e_lj.max())
