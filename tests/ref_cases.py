"""Inputs of the reference-kernel pins, shared by the tests and by the script that records the reference's outputs
(tests/golden/make_ref_pair_kernels.py -> tests/golden/ref_pair_kernels.npz).  Everything here is seeded and deterministic:
the fixture holds OUTPUTS of /root/reference/src/cuda/cuda.cu + util.cu (compiled as they lie into oracle/_ref, run on the
MI355X), the inputs are re-derived from this file."""
from __future__ import annotations

import os

import numpy as np

from molchanica_amd import MdConfig, MdSystem, systems

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pair_kernels.npz")
KE = 332.0637          # mdx_config.coulomb_k default; the reference's kernel has no unit constant (util.cu:53-63)
OVR_BONDED_DISABLED = 0x1
OVR_RECIP_DISABLED = 0x8


# ---- min_image sweep: exact ties on every axis ---------------------------------------------------------------------------
def min_image_cases():
    """(ext, dv) pairs: d = +-L/2, +-3L/2, +-L, +-5L/2 (ties go to even: util.cu:65-71 uses rintf), their fp32 neighbours,
    and random vectors - in a box with three different edges and in dhfr23k's cube."""
    out = []
    rng = np.random.default_rng(77)
    for ext in (np.array([20.0, 30.0, 40.0], np.float32), np.array([62.23, 62.23, 62.23], np.float32),
                np.array([217.238, 217.238, 217.238], np.float32)):
        mult = [0.5, -0.5, 1.5, -1.5, 1.0, -1.0, 2.5, -2.5, 0.0, 0.25, -0.75]
        for axis in range(3):
            for m in mult:
                d = rng.uniform(-0.4, 0.4, 3).astype(np.float32) * ext
                d[axis] = np.float32(m) * ext[axis]
                out.append((ext, d.astype(np.float32)))
                for nb in (np.float32(np.inf), np.float32(-np.inf)):     # one ulp either side of the tie
                    e = d.copy(); e[axis] = np.nextafter(d[axis], nb)
                    out.append((ext, e))
        for m in mult:                                                    # ties on all three axes at once
            out.append((ext, (np.float32(m) * ext).astype(np.float32)))
        for _ in range(40):
            out.append((ext, (rng.uniform(-2.6, 2.6, 3) * ext).astype(np.float32)))
    return out


# ---- the production path of dhfr23k through the reference's arithmetic --------------------------------------------------
def dhfr_case(n_solute=150, n_water=150, seed=41):
    """dhfr23k, rc = 10 A, shifted-cutoff Coulomb: for a few hundred targets the PRE-IMAGED, CUTOFF-FILTERED source set the
    production path sums over (canonical-fp32 distance below the cutoff, exclusions and 1-4 partners removed), as the
    reference's lj_force_kernel / coulomb_force_kernel want it: per target, source positions in the target's frame and
    per-pair sigma / eps.  Built by the oracle's own neighbour search; the reference's kernels then do the arithmetic."""
    from oracle import oracle
    s = systems.dhfr23k()
    cfg = MdConfig(lj_cutoff=10.0, coulomb_cutoff=10.0, skin=2.0, overrides=OVR_BONDED_DISABLED | OVR_RECIP_DISABLED)
    pos = oracle.wrap(s, s.pos)
    rng = np.random.default_rng(seed)
    n_chain = 2489
    targets = np.concatenate([rng.choice(n_chain, n_solute, replace=False), n_chain + rng.choice(s.n_atoms - n_chain, n_water, replace=False)])
    off, idx = oracle.neighbor_list(s, 10.0, pos=pos, use_cells=True)
    excl = [set() for _ in range(s.n_atoms)]
    tset = set(int(t) for t in targets)
    eo, ei = np.asarray(s.excl_offsets), np.asarray(s.excl_idx)
    for t in tset:
        excl[t].update(int(j) for j in ei[eo[t]:eo[t + 1]])
    for a, b in np.asarray(s.pairs14_idx).reshape(-1, 2):
        if int(a) in tset: excl[int(a)].add(int(b))
        if int(b) in tset: excl[int(b)].add(int(a))
    L = (np.asarray(s.box_hi, np.float32) - np.asarray(s.box_lo, np.float32)).astype(np.float32)
    sig, eps = np.asarray(s.lj_sigma, np.float64), np.asarray(s.lj_eps, np.float64)
    cases = []
    for t in targets:
        t = int(t)
        nb = np.array([int(j) for j in idx[off[t]:off[t + 1]] if int(j) not in excl[t]], dtype=np.int64)
        d = (pos[t][None, :] - pos[nb]).astype(np.float32)
        d = (d - np.rint(d / L).astype(np.float32) * L).astype(np.float32)           # the canonical minimum image
        src = (pos[t][None, :] - d).astype(np.float32)                                # sources in the target's frame
        tt, tj = int(s.lj_type[t]), np.asarray(s.lj_type)[nb]
        cases.append(dict(target=t, tgt=pos[t].astype(np.float32), src=src, nb=nb,
                          sigma=(0.5 * (sig[tt] + sig[tj])).astype(np.float32), eps=np.sqrt(eps[tt] * eps[tj]).astype(np.float32),
                          q_t=np.float32(s.charge[t]), q_src=np.asarray(s.charge, np.float32)[nb]))
    return s, cfg, pos, targets, cases


def run_reference_on_dhfr(ref, cases):
    """-> (f_lj [n,3], f_coul [n,3] WITHOUT the unit constant): the reference's kernels on every target's source set."""
    f_lj = np.zeros((len(cases), 3), np.float64); f_c = np.zeros((len(cases), 3), np.float64)
    far = np.array([[1.0e6, 1.0e6, 1.0e6]], np.float32)
    for k, c in enumerate(cases):
        f_lj[k] = ref.lj_force(c["tgt"][None, :], c["src"], c["sigma"][None, :], c["eps"][None, :])[0]
        # coulomb_force_kernel reads ONE charge array for sources and targets: slot 0 is the target's charge, carried by a
        # dummy source a million angstroms away (its own contribution is q_t^2 / 3e12)
        f_c[k] = ref.coulomb_force(c["tgt"][None, :], np.concatenate([far, c["src"]]), np.concatenate([[c["q_t"]], c["q_src"]]))[0]
    return f_lj, f_c


# ---- the formula cases of tests/test_gpu_reference_kernels.py -----------------------------------------------------------
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


def load_fixture():
    return np.load(FIXTURE) if os.path.exists(FIXTURE) else None
