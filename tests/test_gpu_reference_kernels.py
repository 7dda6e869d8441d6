"""Pins the oracle - and the engine - against the REFERENCE'S OWN pair kernels, executed on the MI355X.

The engine the reference calls is the absent crate `dynamics`; the only native code on this path that IS in the tree is
/root/reference/src/cuda/cuda.cu + util.cu (`lj_force_kernel`, `coulomb_force_kernel`, `lj_V_kernel`, `min_image`).  They
are plain CUDA C++ and hipcc compiles them as they lie (oracle/Makefile target `ref` -> oracle/_ref/libref_cuda.so; the .so
travels to the GPU box, the sources do not).  Here the reference's kernels run on random target / source sets and the
oracle's pair terms (LJ 12-6 force and energy, the tgt - src direction, the Coulomb form with its
softening, minimum image by rint) must reproduce them; the engine is held to the same numbers."""
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, MdSystem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_kernels
    if not ref_kernels.available():
        ref_kernels.build()
    if not ref_kernels.available():      # a checkout without the prebuilt object on a machine without the reference tree
        pytest.skip("oracle/_ref/libref_cuda.so is not here and /root/reference is not either: build it in the build container (make -C oracle ref)")
    return ref_kernels


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1
    return md_state


def two_groups(seed, n_a=40, n_b=60, min_dist=2.2):
    """Targets A and sources B: random points in a 16 A cube, no two closer than min_dist."""
    rng = np.random.default_rng(seed)
    pts = []
    while len(pts) < n_a + n_b:
        p = rng.uniform(0, 16, 3)
        if all(np.linalg.norm(p - q) >= min_dist for q in pts):
            pts.append(p)
    pts = np.array(pts, np.float32)
    return pts[:n_a], pts[n_a:], rng


def system(pos, charge, types, sigma, eps):
    n = len(pos)
    return MdSystem(pos=pos, mass=np.full(n, 12.0), charge=charge, lj_type=types, lj_sigma=sigma, lj_eps=eps).normalise()


def forces_on_a_from_b(evaluate, a, b, qa, qb, ta, tb, sigma, eps, cfg):
    """Pairwise additivity: F_A(A u B) - F_A(A alone) is the force the sources exert on the targets."""
    full = system(np.concatenate([a, b]), np.concatenate([qa, qb]), np.concatenate([ta, tb]), sigma, eps)
    alone = system(a, qa, ta, sigma, eps)
    f_full, e_full = evaluate(full, cfg)
    f_alone, e_alone = evaluate(alone, cfg)
    e_b = evaluate(system(b, qb, tb, sigma, eps), cfg)[1]
    return np.asarray(f_full, np.float64)[:len(a)] - np.asarray(f_alone, np.float64), e_full, e_alone, e_b


NOCUT = dict(lj_cutoff=0.0, coulomb_cutoff=0.0)


def test_lj_force_kernel_pins_the_12_6_form_and_direction(ref, orc, mdx):
    a, b, rng = two_groups(1)
    sig_t, eps_t = np.array([3.4, 3.0, 2.6]), np.array([0.10, 0.17, 0.05])
    ta, tb = rng.integers(0, 3, len(a)), rng.integers(0, 3, len(b))
    sigma_ts = 0.5 * (sig_t[ta][:, None] + sig_t[tb][None, :])            # Lorentz-Berthelot table, [n_tgt, n_src]
    eps_ts = np.sqrt(eps_t[ta][:, None] * eps_t[tb][None, :])
    f_ref = ref.lj_force(a, b, sigma_ts, eps_ts).astype(np.float64)
    assert np.abs(f_ref).max() > 1.0
    z_a, z_b = np.zeros(len(a)), np.zeros(len(b))
    cfg = MdConfig(**NOCUT)
    f_orc, *_ = forces_on_a_from_b(lambda s, c: orc.forces(s, c), a, b, z_a, z_b, ta, tb, sig_t, eps_t, cfg)
    scale = np.maximum(np.linalg.norm(f_ref, axis=1), 1.0)
    assert (np.linalg.norm(f_orc - f_ref, axis=1) <= 2e-5 * scale).all(), "oracle LJ force differs from the reference's lj_force_kernel"

    def eng(s, c):
        with mdx.MdState(s, c) as md:
            return md.forces(), md.energy()
    f_gpu, *_ = forces_on_a_from_b(eng, a, b, z_a, z_b, ta, tb, sig_t, eps_t, cfg)
    assert (np.linalg.norm(f_gpu - f_ref, axis=1) <= 1e-4 * scale).all(), "engine LJ force differs from the reference's lj_force_kernel"


def test_coulomb_force_kernel_pins_form_softening_and_sign(ref, orc, mdx):
    """F = dir q_s q_t / (r^2 + 1e-6), dir = (tgt - src)/r, no unit constant (util.cu:9, 53-63): coulomb_k = 1 and
    softening_sq = 1e-6 in this repo's config reproduce it."""
    a, b, rng = two_groups(2, 48, 48)
    q = rng.normal(0, 0.4, 48)
    f_ref = ref.coulomb_force(a, b, q).astype(np.float64)        # source i and target j both read q[.]
    cfg = MdConfig(coulomb_k=1.0, softening_sq=1e-6, overrides=0x4 | 0x8, **NOCUT)     # LJ off
    types = np.zeros(48, int)
    f_orc, *_ = forces_on_a_from_b(lambda s, c: orc.forces(s, c), a, b, q, q, types, types, [3.0], [0.0], cfg)
    scale = np.maximum(np.linalg.norm(f_ref, axis=1), 1e-2)
    assert (np.linalg.norm(f_orc - f_ref, axis=1) <= 2e-5 * scale).all()
    # like charges repel along tgt - src: one pair, checked by hand
    one = ref.coulomb_force([[1.0, 0, 0]], [[0.0, 0, 0]], [0.5])
    assert one[0, 0] == pytest.approx(0.25 / (1.0 + 1e-6), rel=1e-6) and abs(one[0, 1]) == 0.0

    def eng(s, c):
        with mdx.MdState(s, c) as md:
            return md.forces(), md.energy()
    f_gpu, *_ = forces_on_a_from_b(eng, a, b, q, q, types, types, [3.0], [0.0], cfg)
    assert (np.linalg.norm(f_gpu - f_ref, axis=1) <= 1e-4 * scale + 1e-6).all()


def test_lj_V_kernel_pins_the_energy_form(ref, orc, mdx):
    a, b, _ = two_groups(3)
    sigma, eps = 3.2, 0.12
    v_ref = ref.lj_V(b, a, sigma, eps).astype(np.float64)        # per target: energy with all sources
    cfg = MdConfig(**NOCUT)
    z_a, z_b, ta, tb = np.zeros(len(a)), np.zeros(len(b)), np.zeros(len(a), int), np.zeros(len(b), int)
    _, e_full, e_a, e_b = forces_on_a_from_b(lambda s, c: orc.forces(s, c), a, b, z_a, z_b, ta, tb, [sigma], [eps], cfg)
    cross = e_full["lj"] - e_a["lj"] - e_b["lj"]
    assert cross == pytest.approx(v_ref.sum(), rel=2e-5, abs=1e-5)
    assert float(ref.lj_V([[0, 0, 0]], [[2 ** (1 / 6) * sigma, 0, 0]], sigma, eps)[0]) == pytest.approx(-eps, rel=1e-5)   # K1 on the reference itself

    def eng(s, c):
        with mdx.MdState(s, c) as md:
            return md.forces(), md.energy()
    _, g_full, g_a, g_b = forces_on_a_from_b(eng, a, b, z_a, z_b, ta, tb, [sigma], [eps], cfg)
    assert g_full["lj"] - g_a["lj"] - g_b["lj"] == pytest.approx(v_ref.sum(), rel=1e-4, abs=1e-3)


def test_min_image_is_rint_half_even(ref, orc):
    """`min_image` (util.cu:65-71): d - rintf(d / L) L per axis - ties go to even, so d = 0.5 L stays, d = 1.5 L -> -0.5 L.
    The oracle's canonical fp32 distance (and with it the bit-exact neighbour lists) uses exactly this."""
    ext = np.array([20.0, 30.0, 40.0], np.float32)
    for dv in ([12.0, -16.0, 21.0], [10.0, 15.0, 20.0], [30.0, -45.0, 60.0], [0.1, -0.1, 39.9], [-10.0, -15.0, -20.0]):
        got = ref.min_image(ext, dv)
        d = np.array(dv, np.float32)
        want = (d - np.rint(d / ext).astype(np.float32) * ext).astype(np.float32)
        assert np.array_equal(got, want), (dv, got, want)
    # through the oracle: a two-atom periodic system separated by 0.6 L feels the image at -0.4 L (K6)
    s = MdSystem(pos=[[1.0, 1.0, 1.0], [13.0, 1.0, 1.0]], mass=[12, 12], charge=[0.3, -0.3], lj_type=[0, 0], lj_sigma=[3.0], lj_eps=[0.0],
                 periodic=True, box_lo=(0, 0, 0), box_hi=(20.0, 30.0, 40.0)).normalise()
    f, _ = orc.forces(s, MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=0.5, coulomb_k=1.0, overrides=0x4 | 0x8))
    d = ref.min_image(ext, [1.0 - 13.0, 0.0, 0.0])              # tgt - src for atom 0
    assert d[0] == pytest.approx(8.0) and f[0, 0] == pytest.approx(-0.09 / 64.0 * 1.0, rel=1e-5)   # d = +8: atom 0 sits 8 A on the +x side of atom 1's nearest image, and is pulled towards it (-x)
