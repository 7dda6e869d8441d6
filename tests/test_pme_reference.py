"""Pins the numpy SPME restatement (oracle/pme_ref.py) against the textbook reciprocal Ewald sum.  CPU only."""
import math

import numpy as np
import pytest

from oracle import pme_ref as P


def system(n=40, seed=0):
    rng = np.random.default_rng(seed)
    box = np.array([20.0, 22.0, 24.0])
    pos = rng.uniform(0, 1, (n, 3)) * box
    q = rng.normal(size=n)
    q -= q.mean()
    return pos, q, box


def test_direct_sum_force_is_minus_gradient():
    pos, q, box = system()
    e0, f0 = P.ewald_recip_direct(pos, q, box, 0.35)
    h = 1e-5
    for i, a in ((3, 1), (17, 0)):
        p2, p3 = pos.copy(), pos.copy()
        p2[i, a] += h
        p3[i, a] -= h
        fd = -(P.ewald_recip_direct(p2, q, box, 0.35)[0] - P.ewald_recip_direct(p3, q, box, 0.35)[0]) / (2 * h)
        assert fd == pytest.approx(f0[i, a], rel=1e-6)
    assert np.abs(f0.sum(0)).max() < 1e-9


def test_spme_converges_to_the_direct_sum():
    pos, q, box = system()
    e0, f0 = P.ewald_recip_direct(pos, q, box, 0.35)
    rms = lambda f: math.sqrt(((f - f0) ** 2).sum(1).mean()) / math.sqrt((f0 ** 2).sum(1).mean())
    e1, f1 = P.spme_recip(pos, q, (0, 0, 0), box, 0.35, (20, 24, 24), 4)
    e2, f2 = P.spme_recip(pos, q, (0, 0, 0), box, 0.35, (40, 45, 48), 4)
    e3, f3 = P.spme_recip(pos, q, (0, 0, 0), box, 0.35, (40, 45, 48), 6)
    assert abs(e1 - e0) / abs(e0) < 2e-3 and rms(f1) < 1e-2
    assert abs(e2 - e0) / abs(e0) < 1e-4 and rms(f2) < 1e-3
    assert abs(e3 - e0) / abs(e0) < 1e-6 and rms(f3) < 1e-5
    # translation of everything by a lattice-incommensurate vector changes nothing physical
    e4, _ = P.spme_recip(pos + 0.37, q, (0, 0, 0), box, 0.35, (40, 45, 48), 6)
    assert e4 == pytest.approx(e3, rel=1e-6)


def test_madelung_constant_of_rock_salt():
    """Full Ewald energy (real + reciprocal + self) of NaCl: -1.747565 k_e / a per ion pair."""
    from scipy.special import erfc
    a = 2.8
    n = 4
    g = np.arange(n)
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    q = np.where(sites.sum(1) % 2 == 0, 1.0, -1.0)
    pos = sites * a
    box = np.full(3, n * a)
    beta = 0.6
    e_rec, _ = P.spme_recip(pos, q, (0, 0, 0), box, beta, (32, 32, 32), 6)
    d = pos[:, None] - pos[None]
    e_real = 0.0
    for sx in (-1, 0, 1):
        for sy in (-1, 0, 1):
            for sz in (-1, 0, 1):
                dd = d + np.array([sx, sy, sz]) * box
                r = np.linalg.norm(dd, axis=-1)
                m = r > 1e-9
                e_real += 0.5 * P.KE * (np.outer(q, q)[m] * erfc(beta * r[m]) / r[m]).sum()
    e = e_real + e_rec + P.ewald_self_energy(q, beta)
    assert e / (len(q) / 2) == pytest.approx(-1.747565 * P.KE / a, rel=2e-5)


def _stockham(x, inverse=False):
    """The x pass of pme_xpass_solve_kernel (molchanica_amd/csrc/mdx_pme.hip) restated: autosort Stockham passes of radix 4, 2, 3, 5,
    y[q + s (r p + u)] = w_N^(p u s) sum_t x[q + s (p + t m)] w_r^(t u), n -> n / r, s -> s r; unnormalised both ways like hipFFT."""
    N = len(x)
    fac, rest = [], N
    for r in (4, 2, 3, 5):
        while rest % r == 0 and not (r == 2 and rest % 4 == 0):
            fac.append(r); rest //= r
    assert rest == 1, "mesh sizes are 2-3-5-smooth (good_size)"
    tw = np.exp(-2j * np.pi * np.arange(N) / N)
    if inverse:
        tw = np.conj(tw)
    a, b = x.astype(np.complex128).copy(), np.zeros(N, np.complex128)
    n, s = N, 1
    for r in fac:
        m = n // r
        for bf in range(N // r):
            p, q = divmod(bf, s)
            v = [a[q + s * (p + t * m)] for t in range(r)]
            for u in range(r):
                assert p * u * s < N                      # the kernel reads the twiddle table without a modulo
                b[q + s * (r * p + u)] = tw[p * u * s] * sum(v[t] * tw[((N // r) * t * u) % N] for t in range(r))
        a, b = b, a
        n, s = m, s * r
    return a


@pytest.mark.parametrize("n", [8, 12, 20, 27, 30, 50, 64, 96, 100, 200, 240, 250, 256, 384, 512])
def test_stockham_x_pass_equals_the_dft(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.abs(_stockham(x) - np.fft.fft(x)).max() < 1e-11 * n
    assert np.abs(_stockham(x, inverse=True) - np.fft.ifft(x) * n).max() < 1e-11 * n
