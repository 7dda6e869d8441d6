"""Reciprocal-space Ewald, two independent ways (TEST INFRASTRUCTURE ONLY, numpy fp64):

  ewald_recip_direct  — the textbook reciprocal sum over explicit k vectors (exact up to k_max)
  spme_recip          — smooth particle-mesh Ewald (Essmann et al. 1995), cubic B-splines by default,
                        the same conventions as molchanica_amd/csrc/mdx_pme.hip

plus the corrections that go with them (self term, excluded pairs, neutralising background).
The reference's Coulomb is SPME unless `long_range_recip_disabled` (/root/reference README.md:240,
src/mol_editor/mod.rs:873); its implementation lives in the absent `ewald` crate."""
from __future__ import annotations

import math

import numpy as np

KE = 332.0637


def ewald_recip_direct(pos, q, box, beta, kmax=None, ke=KE):
    """-> (energy, forces [N,3]).  E = (2 pi ke / V) sum_{k != 0} exp(-k^2/4 beta^2)/k^2 |S(k)|^2."""
    pos, q, box = np.asarray(pos, float), np.asarray(q, float), np.asarray(box, float)
    v = box.prod()
    if kmax is None:
        kmax = [int(math.ceil(beta * L * math.sqrt(-math.log(1e-14)) / math.pi)) for L in box]
    n = [np.arange(-k, k + 1) for k in kmax]
    g = np.stack(np.meshgrid(*n, indexing="ij"), -1).reshape(-1, 3)
    g = g[(g != 0).any(1)]
    kv = 2 * math.pi * g / box
    k2 = (kv ** 2).sum(1)
    a = np.exp(-k2 / (4 * beta * beta)) / k2
    keep = a > 1e-16 * a.max()
    kv, a = kv[keep], a[keep]
    e, f = 0.0, np.zeros_like(pos)
    for s in range(0, kv.shape[0], 4096):
        kk, aa = kv[s:s + 4096], a[s:s + 4096]
        ph = pos @ kk.T                                   # [N, nk]
        c, sn = np.cos(ph), np.sin(ph)
        sr, si = q @ c, q @ sn                            # S(k) = sum q e^{ik.r}
        e += (aa * (sr * sr + si * si)).sum()
        # F_i = (4 pi ke q_i / V) sum_k a(k) k [ sin(k.r_i) Re S - cos(k.r_i) Im S ]
        w = aa * (sn * sr - c * si)                       # [N, nk]
        f += (w @ kk) * q[:, None]
    return 2 * math.pi * ke / v * e, 4 * math.pi * ke / v * f


def _bspline(order, w):
    """Values and derivatives of M_n(w + j), j = 0..n-1, for fractional offsets w in [0,1).  [n, len(w)]."""
    w = np.asarray(w, float)
    m = np.zeros((order, w.size))
    m[0] = 1.0 - w
    m[1] = w
    for k in range(3, order):                              # build M_{k} from M_{k-1}
        div = 1.0 / (k - 1)
        m[k - 1] = div * w * m[k - 2]
        for j in range(1, k - 1):
            m[k - 1 - j] = div * ((w + j) * m[k - 2 - j] + (k - j - w) * m[k - 1 - j])
        m[0] = div * (1 - w) * m[0]
    # derivative from order n-1 values, then the last recursion to order n
    d = np.zeros_like(m)
    d[0] = -m[0]
    for j in range(1, order):
        d[j] = m[j - 1] - m[j]
    k = order
    div = 1.0 / (k - 1)
    m[k - 1] = div * w * m[k - 2]
    for j in range(1, k - 1):
        m[k - 1 - j] = div * ((w + j) * m[k - 2 - j] + (k - j - w) * m[k - 1 - j])
    m[0] = div * (1 - w) * m[0]
    # m[j] now holds M_n evaluated at w + (n-1-j)?  normalise the convention below in spme_recip
    return m, d


def bspline_moduli(K, order):
    """|b(m)|^2 for m = 0..K-1 (Essmann eq. 4.4)."""
    mvals, _ = _bspline(order, np.array([0.0]))
    mn = mvals[:, 0]                                       # M_n at the integer knots
    k = np.arange(order)
    mm = np.arange(K)
    den = (mn[None, :] * np.exp(2j * math.pi * mm[:, None] * k[None, :] / K)).sum(1)
    b2 = 1.0 / np.maximum(np.abs(den) ** 2, 1e-30)
    bad = np.abs(den) ** 2 < 1e-7                          # only for even order at m = K/2
    if bad.any():
        for i in np.nonzero(bad)[0]:
            b2[i] = 0.5 * (b2[i - 1] + b2[(i + 1) % K])
    return b2


def theta_table(grid, box, beta, order):
    """theta(m) = B(m) exp(-pi^2 m~^2/beta^2) / (pi V m~^2), full [K1,K2,K3] table, theta(0) = 0."""
    box = np.asarray(box, float)
    v = box.prod()
    ms = []
    for K, L in zip(grid, box):
        m = np.arange(K)
        m = np.where(m <= K // 2, m, m - K)
        ms.append(m / L)
    m2 = ms[0][:, None, None] ** 2 + ms[1][None, :, None] ** 2 + ms[2][None, None, :] ** 2
    b = (bspline_moduli(grid[0], order)[:, None, None] * bspline_moduli(grid[1], order)[None, :, None]
         * bspline_moduli(grid[2], order)[None, None, :])
    with np.errstate(divide="ignore", invalid="ignore"):
        th = b * np.exp(-math.pi ** 2 * m2 / beta ** 2) / (math.pi * v * m2)
    th[0, 0, 0] = 0.0
    return th


def spme_recip(pos, q, box_lo, box, beta, grid, order=4, ke=KE):
    """-> (energy, forces).  Q spread with M_n, E = 1/2 sum theta |F(Q)|^2, F_i = -sum dQ_i/dr phi."""
    pos, q, box = np.asarray(pos, float), np.asarray(q, float), np.asarray(box, float)
    n = pos.shape[0]
    grid = list(grid)
    u = (pos - np.asarray(box_lo, float)) / box
    u = (u - np.floor(u)) * np.asarray(grid)
    fl = np.floor(u).astype(int)
    w = u - fl
    wts, dws, idx = [], [], []
    for d in range(3):
        m, dm = _bspline(order, w[:, d])                   # m[j]: weight of grid point fl - (order-1) + j ... see below
        wts.append(m)
        dws.append(dm)
        # M_n(u - k) is non-zero for k = fl - (n-1) .. fl ; _bspline's row j belongs to k = fl - (n-1) + j
        idx.append((fl[:, d][None, :] - (order - 1) + np.arange(order)[:, None]) % grid[d])
    Q = np.zeros(grid)
    for a in range(order):
        for b in range(order):
            for c in range(order):
                np.add.at(Q, (idx[0][a], idx[1][b], idx[2][c]), q * wts[0][a] * wts[1][b] * wts[2][c])
    th = theta_table(grid, box, beta, order)
    FQ = np.fft.fftn(Q)
    e = 0.5 * ke * float((th * (FQ.real ** 2 + FQ.imag ** 2)).sum())
    phi = np.fft.ifftn(th * FQ).real * Q.size               # unnormalised inverse
    f = np.zeros((n, 3))
    scale = np.asarray(grid) / box
    for a in range(order):
        for b in range(order):
            for c in range(order):
                p = phi[idx[0][a], idx[1][b], idx[2][c]]
                f[:, 0] -= q * dws[0][a] * wts[1][b] * wts[2][c] * p * scale[0]
                f[:, 1] -= q * wts[0][a] * dws[1][b] * wts[2][c] * p * scale[1]
                f[:, 2] -= q * wts[0][a] * wts[1][b] * dws[2][c] * p * scale[2]
    return e, ke * f


def spme_recip_virial(pos, q, box_lo, box, beta, grid, order=4, ke=KE):
    """Scalar virial W = -dE/dlambda of the SPME reciprocal energy under r -> lambda r, L -> lambda L:
    sum_m E_m (1 - 2 pi^2 m^2 / beta^2)  (S(m) and the B-spline moduli are scale invariant)."""
    pos, q, box = np.asarray(pos, float), np.asarray(q, float), np.asarray(box, float)
    grid = list(grid)
    u = (pos - np.asarray(box_lo, float)) / box
    u = (u - np.floor(u)) * np.asarray(grid)
    fl = np.floor(u).astype(int)
    w = u - fl
    wts, idx = [], []
    for d in range(3):
        m, _ = _bspline(order, w[:, d])
        wts.append(m)
        idx.append((fl[:, d][None, :] - (order - 1) + np.arange(order)[:, None]) % grid[d])
    Q = np.zeros(grid)
    for a in range(order):
        for b in range(order):
            for c in range(order):
                np.add.at(Q, (idx[0][a], idx[1][b], idx[2][c]), q * wts[0][a] * wts[1][b] * wts[2][c])
    th = theta_table(grid, box, beta, order)
    FQ = np.fft.fftn(Q)
    ms = []
    for K, L in zip(grid, box):
        m = np.arange(K)
        ms.append(np.where(m <= K // 2, m, m - K) / L)
    m2 = ms[0][:, None, None] ** 2 + ms[1][None, :, None] ** 2 + ms[2][None, None, :] ** 2
    em = 0.5 * ke * th * (FQ.real ** 2 + FQ.imag ** 2)
    return float((em * (1.0 - 2.0 * math.pi ** 2 * m2 / beta ** 2)).sum())


def ewald_self_energy(q, beta, ke=KE):
    return -ke * beta / math.sqrt(math.pi) * float((np.asarray(q, float) ** 2).sum())


def ewald_background_energy(q, box, beta, ke=KE):
    qt = float(np.asarray(q, float).sum())
    return -math.pi * ke * qt * qt / (2.0 * float(np.prod(box)) * beta * beta)


def excluded_pair_correction(pos, q, pairs, box, beta, ke=KE):
    """Remove erf(beta r)/r for pairs the real-space sum skips.  -> (energy, forces)."""
    from scipy.special import erf
    pos, q, box = np.asarray(pos, float), np.asarray(q, float), np.asarray(box, float)
    f = np.zeros_like(pos)
    pairs = np.asarray(pairs, int).reshape(-1, 2)
    if pairs.shape[0] == 0:
        return 0.0, f
    d = pos[pairs[:, 0]] - pos[pairs[:, 1]]
    d -= np.round(d / box) * box
    r = np.linalg.norm(d, axis=1)
    kqq = ke * q[pairs[:, 0]] * q[pairs[:, 1]]
    e = -(kqq * erf(beta * r) / r).sum()
    fs = -kqq * (erf(beta * r) / r ** 3 - 2 * beta / math.sqrt(math.pi) * np.exp(-(beta * r) ** 2) / r ** 2)
    np.add.at(f, pairs[:, 0], fs[:, None] * d)
    np.add.at(f, pairs[:, 1], -fs[:, None] * d)
    return float(e), f
