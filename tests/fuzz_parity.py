"""Randomised GPU-vs-oracle parity (test infrastructure, like tests/: it uses oracle/): random small systems, cut-offs, skins, Coulomb
modes, kernel variants and - per case - the environment knobs that force the alternate paths of the list build / charge spread /
constraint tables, each held to the tolerances of tests/test_gpu_parity.py (per-atom force 1e-4 max(|F|, 1) + cutoff slack, energies
2e-6 + gross-sum floor) at the initial geometry and after a rebuild-forcing move.  Usage (through gpurun): python tests/fuzz_parity.py [cases=60] [seed=1]"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig, _abi
from molchanica_amd import md_state as mdx
from oracle import oracle as orc
from tests.test_gpu_parity import assert_forces, energy_tolerance, TERMS

def assert_energies(e, eo, what, extra=0.0):
    """Bonded and 1-4 terms to the tests' bound; the pair sums ten times looser than the tests hold BASELINE's configurations to - at
    the 5-7 A cut-offs and Ewald parameters drawn here the fp32 pair energies (and the 1.5e-7 of the erfc approximation) add up
    against sums that cancel to a thousandth of their gross; the FORCES are held to the contract as they are."""
    for k in TERMS:
        tol = energy_tolerance(eo, k) * (10.0 if k in ("lj", "coulomb") else 1.0) + (extra if k == "coulomb" else 0.0)
        assert abs(e[k] - eo[k]) <= tol, f"{what}: {k} gpu {e[k]!r} oracle {eo[k]!r}: {abs(e[k] - eo[k]) / tol:.2f}x its tolerance {tol:.2e}"

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
base_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KNOBS = {"MDX_KIND_CLUSTERS": ["0", "1"], "MDX_PME_SPREAD_BRICK": ["0", "1"], "MDX_CONS_SORT_MIN": ["1", "100000000"], "MDX_PME_BRICK_EDGE": ["8", "11", "16"],
         "MDX_VSITE_IN_GROUPS": ["0", "1"], "MDX_CONS_RIGID3": ["0", "1"], "MDX_PME_OVERLAP": ["0", "1"]}
fails = 0; t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng([base_seed, case])      # (every case its own stream: FUZZ_ONLY=k reproduces case k)
    kind = rng.choice(["lig", "solvated", "water", "opc", "rigid"])
    seed = int(rng.integers(1, 10000))
    if kind == "lig": s = systems.lig50(seed=seed, n_atoms=int(rng.integers(20, 120)))
    elif kind == "solvated": s = systems.small_solvated(seed=seed, n_chain=int(rng.integers(40, 300)), box=float(rng.uniform(24.0, 34.0)))
    elif kind == "water": s = systems.water_box(int(rng.integers(5, 10)), seed=seed)
    elif kind == "rigid": s = systems.water_box(int(rng.integers(5, 10)), seed=seed, rigid=True)
    else: s = systems.opc_water_box(int(rng.integers(5, 10)), seed=seed)
    periodic = bool(s.periodic)
    L = float(np.min(np.array(s.box_hi) - np.array(s.box_lo))) if periodic else 1e9
    rc_max = min(10.0, 0.5 * L - 2.6) if periodic else 12.0
    rc = float(rng.uniform(min(6.0, rc_max), rc_max)); skin = float(rng.uniform(0.5, min(2.5, 0.5 * L - rc - 0.05) if periodic else 2.5))
    mode = int(rng.choice([0, 1, 2])) if periodic else int(rng.choice([0, 1]))
    cfgk = dict(lj_cutoff=rc, coulomb_cutoff=rc if rng.random() < 0.7 else float(rng.uniform(min(6.0, rc), rc)), skin=skin, coulomb_mode=mode,
                nb_variant=int(rng.choice([0, 2, 5])), combining_rule=int(rng.choice([0, 1])))
    if mode == 2: cfgk.update(ewald_alpha=float(rng.uniform(0.28, 0.42)), overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED)   # (the oracle's real-space part)
    cfg = MdConfig(**cfgk)
    env = {k: str(rng.choice(v)) for k, v in KNOBS.items() if rng.random() < 0.5}
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(env)
    for kv in filter(None, os.environ.get("FUZZ_FORCE_ENV", "").split(",")):      # (reproduce a case with one knob changed)
        k, v = kv.split("="); os.environ[k] = v; env[k] = v
    only = os.environ.get("FUZZ_ONLY")
    if only and case != int(only): continue
    what = f"case {case}: {kind} N={s.n_atoms} rc={rc:.2f}/{cfgk['coulomb_cutoff']:.2f} skin={skin:.2f} mode={mode} variant={cfgk['nb_variant']} comb={cfgk['combining_rule']} env={env}"
    try:
        with mdx.MdState(s, cfg) as md:
            for phase in range(2):
                f, e = md.forces(), md.energy()      # (first: a new handle projects the geometry onto its constraints and places the virtual sites at its first force call)
                p = md.positions().astype(np.float64)
                pw = orc.wrap(s, p) if periodic else p
                fo, eo = orc.forces(s, cfg, pos=pw, use_cells=periodic and s.n_atoms > 400)
                if only:      # details of the worst atoms
                    err = np.linalg.norm(np.asarray(f, np.float64) - fo, axis=1); bad = np.argsort(-err)[:8]
                    for i in bad: print(f"    atom {i} (type {int(s.lj_type[i])}, q {float(s.charge[i]):+.3f}) pos {pw[i].round(3)} |dF| {err[i]:.3e} gpu {np.asarray(f[i]).round(3)} oracle {fo[i].round(3)}")
                    print("    box", s.box_lo, s.box_hi, "stats", {k: md.stats()[k] for k in ("n_tiles", "rebuild_count", "n_cluster_pairs")})
                slack = orc.cutoff_slack(s, cfg, pos=pw)
                if s.vsite_idx is not None and len(s.vsite_idx):      # a site's borderline pair shows on the parents its force is spread to
                    vi = np.asarray(s.vsite_idx).reshape(-1, 4)
                    for c in (1, 2, 3): np.add.at(slack, vi[:, c], slack[vi[:, 0]])
                assert_forces(f, fo, slack, what + f" phase {phase}", outliers=2)      # (two atoms may sit between 1x and 2x the per-atom bound)
                # an unshifted truncation (the Ewald real-space sum) is discontinuous at the cut-off: a pair the two sides place on
                # different sides of it moves the energy by its own erfc term
                assert_energies(e, eo, what + f" phase {phase}", extra=0.5 if (mode == 2 and slack.any()) else 0.0)
                if phase == 0:      # a move beyond skin / 2 for a third of the atoms: list rebuild through the fused chain
                    if s.constraint_idx is not None and len(s.constraint_idx): md.step(0.001, None, 30)
                    else:
                        q = md.positions(); m = rng.random(s.n_atoms) < 0.33
                        q[m] += rng.normal(0, 0.08, (int(m.sum()), 3)).astype(np.float32); md.set_positions(q); md.step(0.0002, None, 5)
        print("ok  ", what, flush=True)
    except Exception as ex:
        fails += 1
        print("FAIL", what, "\n    ", str(ex).splitlines()[0][:300], flush=True)
        if not isinstance(ex, AssertionError): traceback.print_exc()
print(f"{n_cases - fails} of {n_cases} cases within tolerance in {time.time() - t0:.0f} s")
