"""What the parity tests' tolerances leave in hand, measured: for BASELINE configs C1-C5 the worst per-atom force error as a
multiple of SURVEY 8(c)'s 1e-4 * max(|F|, 1) (no RMS floor), the RMS ratio, the relative error of every energy term, and the
100-step trajectory deviation of dhfr23k (C2) under reaction field and under the shifted cutoff.  The numbers the tests quote
come from here.  Usage (through gpurun): python tests/parity_margins.py [c1 c2 c3 c4 c5 traj]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
from oracle import oracle as orc

TERMS = ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14")
NOCUT = dict(lj_cutoff=0.0, coulomb_cutoff=0.0)
want = set(sys.argv[1:]) or {"c1", "c2", "c3", "c4", "c5", "traj"}


def explain_worst(s, cfg, pos, w, dF, F):
    """Where does the worst atom's error come from?  Its pair forces one by one (numpy, fp64, shifted cutoff + LJ as the kernel
    evaluates them): the GROSS sum G of their magnitudes - what fp32 accumulation rounds - against the net force that the bound
    1e-4 max(|F|, 1) is written in, and whether a pair sits within a wider band of the cutoff than the slack already covers."""
    L = np.asarray(s.box_hi, np.float64) - np.asarray(s.box_lo, np.float64)
    x = pos.astype(np.float64)
    d = x[w] - x
    d -= np.round(d / L) * L
    r2 = (d ** 2).sum(1)
    rc = max(cfg.lj_cutoff, cfg.coulomb_cutoff)
    m = (r2 < rc * rc) & (r2 > 0)
    if s.excl_offsets is not None:
        ex = s.excl_idx[s.excl_offsets[w]:s.excl_offsets[w + 1]]
        m[ex] = False
    j = np.nonzero(m)[0]
    r = np.sqrt(r2[j])
    qq = float(cfg.coulomb_k) * float(s.charge[w]) * s.charge[j].astype(np.float64)
    sig = 0.5 * (s.lj_sigma[s.lj_type[w]] + s.lj_sigma[s.lj_type[j]]).astype(np.float64)
    eps = np.sqrt(float(s.lj_eps[s.lj_type[w]]) * s.lj_eps[s.lj_type[j]].astype(np.float64))
    s6 = (sig / r) ** 6
    fmag = np.abs(qq / r2[j]) + np.abs(24.0 * eps * (2.0 * s6 * s6 - s6) / r)
    G = float(fmag.sum())
    near = np.abs(r2[j] / (rc * rc) - 1.0)
    print(f"    worst atom {w}: {len(j)} partners inside rc, gross sum of |pair force| G = {G:.1f} kcal/mol/A against a net |F| = {F:.3f}: "
          f"|dF| = {dF:.2e} = {dF / G:.1e} of G (an fp32 ulp is 6e-8; ~{len(j)} additions in an order the two sides do not share), "
          f"largest single pair force {fmag.max():.1f}; closest pair to the cutoff at {near.min():.1e} relative in r^2 "
          f"({int((near < 1e-4).sum())} within 1e-4, {int((near < 4e-5).sum())} within the slack's 4e-5)", flush=True)


def single_point(name, s, cfg, use_cells, rel=1e-5):
    with MdState(s, cfg) as md:
        pos = md.positions(); f = md.forces().astype(np.float64); e = md.energy()
    fo, eo = orc.forces(s, cfg, pos=pos.astype(np.float64), use_cells=use_cells)
    slack = orc.cutoff_slack(s, cfg, pos=pos, rel=rel) if s.periodic else np.zeros(s.n_atoms)
    err = np.linalg.norm(f - fo, axis=1)
    fn = np.linalg.norm(fo, axis=1)
    tol = 1e-4 * np.maximum(fn, 1.0) + slack
    ratio = err / tol
    clean = slack == 0
    rms = math.sqrt(np.mean(err[clean] ** 2)) / math.sqrt(np.mean((fo[clean] ** 2).sum(1)))
    f_rms = math.sqrt(np.mean((fo ** 2).sum(1)))
    w = int(np.argmax(ratio))
    print(f"{name}: N {s.n_atoms}  worst |dF| / (1e-4 max(|F|,1) + cutoff slack) = {ratio.max():.3f} (atom {w}: |F| {fn[w]:.3f}, |dF| {err[w]:.2e}; "
          f"atoms above 1.0: {int((ratio > 1).sum())}, above 0.5: {int((ratio > 0.5).sum())})  RMS ratio {rms:.2e} (limit 2e-5)  F_rms {f_rms:.2f}  "
          f"max |dF| / F_rms {err.max() / f_rms:.2e}", flush=True)
    if ratio.max() > 0.9 and s.periodic:
        explain_worst(s, cfg, pos, w, err[w], fn[w])
    out = []
    for k in TERMS:
        d = abs(e[k] - eo[k])
        g = eo.get("gross_" + k, 0.0)
        tol = max(1e-3, 2e-6 * abs(eo[k]) + {"lj": 1e-6, "coulomb": 1e-8}.get(k, 0.0) * g)
        out.append(f"{k} {d / max(abs(eo[k]), 1e-30):.1e} (abs {d:.1e} of {eo[k]:.6g}" + (f", gross {g:.4g}: {d / g:.1e} of it" if g else "") + f"; {d / tol:.2f} of the test's tolerance)")
    print("    energy errors: " + "; ".join(out), flush=True)


if "c1" in want: single_point("C1 lig50", systems.lig50(), MdConfig(**NOCUT), False)
if "c2" in want:
    single_point("C2 dhfr23k", systems.dhfr23k(), MdConfig(), True)
    single_point("small_solvated rc9", systems.small_solvated(), MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5), False)
if "c3" in want: single_point("C3 complex50k", systems.complex50k(), MdConfig(), True)
if "c4" in want: single_point("C4 dna100k", systems.dna100k(), MdConfig(), True)
if "c5" in want: single_point("C5 water1M", systems.water1m(), MdConfig(), True, rel=4e-5)
if "traj" in want:
    s = systems.dhfr23k()
    L = np.array(s.box_hi) - np.array(s.box_lo)
    for mode in (1, 0):
        cfg = MdConfig(coulomb_mode=mode)
        with MdState(s, cfg) as md:
            x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
            md.step(0.0005, None, 100)
            xg, vg = md.positions().astype(np.float64), md.velocities().astype(np.float64)
            rb = md.stats()["rebuild_count"]
        t = time.time()
        xo, vo, _ = orc.step(s, cfg, 0.0005, 100, pos=x0, vel=v0, use_cells=True)
        d = xg - xo; d -= np.round(d / L) * L
        per = np.sqrt((d ** 2).sum(1))
        print(f"C2 dhfr23k 100-step trajectory, coulomb_mode {mode}: RMS deviation {math.sqrt((per ** 2).mean()):.2e} A, max {per.max():.2e} A, "
              f"velocity RMS {math.sqrt(((vg - vo) ** 2).sum(1).mean()):.2e} A/ps, {rb} list builds, oracle {time.time() - t:.0f} s", flush=True)
