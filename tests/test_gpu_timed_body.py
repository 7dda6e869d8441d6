"""The pair-kernel body `bench.py` times, held against the oracle directly.

`MdState::step` is the path (/root/reference src/md/mod.rs:716,748).  At water1M the step loop launches
`nb_cluster_kernel<false, CM_SHIFTED, false, true, 1, true, false, 3, ...>`: ONE wave per tile (>= 12,000 tiles), half list,
merged dual-list body (inner-list walk / pruning pass picked on the device) - an instantiation no small test reaches, because one
wave per tile is chosen by tile count and the dual-list body only runs for force calls of the step loop.  Here:

  * water1M itself: 20 steps beside the oracle's own 20-step trajectory (positions, velocities, every energy term), then on
    across a list rebuild and several pruning passes, and the forces THE STEP LOOP LEFT BEHIND against one oracle evaluation at the
    downloaded positions - SURVEY 8(c)'s per-atom bound, all 1,029,000 atoms;
  * the same for the two-waves class (273 k atoms) and the four-waves class (165 k atoms);
  * MDX_WPT = 1 / 2 / 4 on a 12 k-atom box in child processes (the knob is read once per process), so that every
    waves-per-tile x merged-dual instantiation meets the oracle on a system the oracle finishes in a blink.

`mdx_pair_launch_info` names the instantiation that went out, so a change of the selection rule cannot silently move these tests
onto another body.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

from tests.test_gpu_parity import assert_energies, assert_forces

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1, "no GPU: the HIP path must run here, there is no fallback"
    return md_state


def _rms(d):
    return math.sqrt((d ** 2).sum(1).mean())


def step_loop_forces_vs_oracle(md, orc, s, cfg, what, slack_rel=1e-5, outliers=0):
    """Forces the step loop left behind (dual list, inner masks) against the oracle at the same positions; then the plain-list
    evaluation of the same positions (mdx_energy) against both."""
    pos = md.positions()
    f_step = md.forces().astype(np.float64)                 # NOT re-evaluated: what the last step's pair launch produced
    fo, eo = orc.forces(s, cfg, pos=pos.astype(np.float64), use_cells=True)
    slack = orc.cutoff_slack(s, cfg, pos=pos, rel=slack_rel)
    assert_forces(f_step, fo, slack, what + " step-loop forces", outliers=outliers)
    e = md.energy()                                         # plain list, energy flavour, same positions
    assert_energies(e, eo, what + " energies after the steps")
    assert_forces(md.forces(), fo, slack, what + " plain-list forces", outliers=outliers)
    return e, eo


def test_c5_water1m_step_loop_forces_against_the_oracle(mdx, orc):
    s = systems.water1m()
    cfg = MdConfig()                                         # rc 10, skin 2, inner skin 0.5, shifted cutoff: the bench's
    dt = 0.0005
    with mdx.MdState(s, cfg) as md:
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        md.step(dt, None, 20)
        info = md.pair_launch_info()["step"]
        # the instantiation BENCH_r05.json names: one wave per tile, half list, merged dual-list body, shifted cutoff, force flavour
        # (round 6: dual 5 = the same merged body with the previous step's kick + drift and the bonded roles inside - one launch per step)
        dual = 5 if os.environ.get("MDX_ONEPASS", "1") == "2" else 3      # (the one-wave class takes one launch per step only under MDX_ONEPASS=2: measured slower)
        assert (info["waves_per_tile"], info["dual"], info["half"], info["coulomb"], info["energy"], info["workgroups_per_tile"]) == (1, dual, 1, 0, 0, 1), info
        assert (md.pair_launch_info()["one_launch_steps"] >= 15) == (dual == 5)
        assert info["tiles"] >= 12000
        xg, vg = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        # (a) the oracle's own 20 steps from the same state (fp64 state, cell search)
        xo, vo, eo20 = orc.step(s, cfg, dt, 20, pos=x0, vel=v0, use_cells=True)
        L = np.asarray(s.box_hi, np.float64) - np.asarray(s.box_lo, np.float64)
        d = xg - xo
        d -= np.round(d / L) * L
        rms, vrms = _rms(d), _rms(vg - vo)
        e20 = md.energy()
        ke_o = orc.kinetic(s, vo)
        # Bounds, each with what was measured on the MI355X (round 6) behind it.  fp32 coordinates in a 217 A box carry an ulp of
        # 1.5e-5 A: rounding the drift alone is a random walk of ~3e-5 A over 20 steps.  And the shifted cutoff's force jumps at rc:
        # about 6e-4 pairs per atom and step sit within that rounding of the cutoff, the side such a pair falls on differs between
        # fp32 and fp64 state, and each flip is a kick of ~0.3 A/ps on a hydrogen - that, not the pair kernel's arithmetic, is the
        # velocity deviation (C2's 100-step test measures 0.20 A/ps for the same reason).  Measured: 9.0e-5 A, 2.5e-2 A/ps.
        report = {"pos_rms_A": (rms, 3e-4), "vel_rms_A_per_ps": (vrms, 6e-2)}
        # per term against the oracle's energies of ITS final state: the two states differ by the trajectory deviation above, which
        # moves a sum of N terms by ~ sqrt(N) |F| rms (3 kcal/mol here) whatever computes it: 5e-6 |E| + 1e-6 gross (the single-point
        # tolerance at identical positions is 2e-6 |E|).  Measured: bond 0.31, angle 0.55 (3e-6 of it), lj 0.06, coulomb 2.8 kcal/mol.
        for k in ("bond", "angle", "lj", "coulomb"):
            report[k] = (abs(e20[k] - eo20[k]), max(1e-3, 5e-6 * abs(eo20[k]) + 1e-6 * eo20.get("gross_" + k, abs(eo20[k]))))
        report["kinetic"] = (abs(e20["kinetic"] - ke_o), 2e-6 * ke_o)
        report["total"] = (abs(e20["potential"] + e20["kinetic"] - eo20["potential"] - ke_o), 2e-6 * (abs(eo20["potential"]) + ke_o))
        line = ", ".join(f"{k} {v:.3e} (bound {b:.1e})" for k, (v, b) in report.items())
        print("water1M, 20 steps beside the oracle: " + line)
        # (b) on, across a rebuild and more pruning passes; the forces the step loop left behind against the oracle
        md.step(dt, None, 12)
        st = md.stats()
        assert st["rebuild_count"] >= 2 and st["prune_passes"] >= 3, (st["rebuild_count"], st["prune_passes"])
        assert st["rebuild_fallbacks"] == 0
        # round 6: at this size the rebuild's own pruning pass writes the inner list; the force call behind it walks it
        assert md.pair_launch_info()["inner_lists_from_rebuilds"] >= 1
        info2 = md.pair_launch_info()["step"]
        assert {k: v for k, v in info2.items() if k != "tiles"} == {k: v for k, v in info.items() if k != "tiles"}, (info, info2)
        # (outliers: the box as generated is at ~1300 K by now; measured (tools/dbg/c5_outliers.py): three atoms of 1,029,000 between 1 x and
        # 1.61 x the bound, all three oxygens with a net |F| < 3 kcal/mol/A - bound 1e-4 - whose ~420 pair forces add up to a gross ~2000:
        # |dF| = 2e-4 ... 4e-4 is 1-2 fp32 ulp of what is being summed, in an order the two sides do not share; the plain-list evaluation
        # of the same positions shows the same three atoms.  No pair within 4e-5 of the cutoff is involved (those carry their slack).
        # Five in a million may sit between 1 x and 2 x, none beyond.)
        step_loop_forces_vs_oracle(md, orc, s, cfg, "water1M after 32 steps", slack_rel=4e-5, outliers=5)
    assert all(v <= b for v, b in report.values()), line


@pytest.mark.parametrize("n_side,waves", [(45, 2), (38, 4)])
def test_mid_size_classes_step_loop_forces_against_the_oracle(mdx, orc, n_side, waves):
    """273 k atoms (~4300 tiles: two waves per tile) and 165 k atoms (~2600 tiles: four), merged dual-list body."""
    s = systems.water_box(n_side, seed=6)
    cfg = MdConfig()
    with mdx.MdState(s, cfg) as md:
        md.step(0.0005, None, 27)
        info = md.pair_launch_info()["step"]
        assert (info["waves_per_tile"], info["dual"], info["half"]) == (waves, 3, 1), info
        st = md.stats()
        assert st["prune_passes"] >= 3
        step_loop_forces_vs_oracle(md, orc, s, cfg, f"water_box({n_side})", slack_rel=2e-5)
        md.step(0.0005, None, 30)
        assert md.stats()["rebuild_count"] >= 2
        step_loop_forces_vs_oracle(md, orc, s, cfg, f"water_box({n_side}) after a rebuild", slack_rel=2e-5)


@pytest.mark.parametrize("wpt,fused", [(1, False), (2, False), (4, False), (8, False), (1, True), (2, True), (1, "inner"), (1, "onepass"), (2, "generic")])
def test_every_waves_per_tile_instantiation_against_the_oracle(wpt, fused):
    """MDX_WPT is read once per process: each value in a child (tests/timed_body_child.py).  `fused`: MDX_WPT8_BELOW=32 also
    selects the large classes' fused bonded + kick + drift pass - with MDX_WPT=1 the complete water1M arrangement on 12 k atoms."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("MDX_")}
    env["MDX_WPT"] = str(wpt)
    if fused:
        env["MDX_WPT8_BELOW"] = "32"
    if fused == "generic":    # ... the fused pass's flavour WITH the dihedral / 1-4 / exclusion branches (a box of water takes the one without by default)
        env["MDX_FUSED_NODIH"] = "0"
    if fused == "onepass":    # ... and one launch per step in the one-wave class (body 5; off by default there: slower at 1 M atoms)
        env["MDX_ONEPASS"] = "2"
    if fused == "inner":      # ... and the one-wave pruning pass of the list rebuild, which then writes the inner list itself (water1M's path)
        env["MDX_PRUNE_MW_BELOW"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "timed_body_child.py"), str(wpt)] + ([str(fused).replace("True", "fused")] if fused else []),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-25:])
    assert p.returncode == 0 and "TIMED-BODY-OK" in p.stdout, tail


def test_default_operating_point_step_loop_forces_against_the_oracle(mdx, orc):
    """The body `bench.py`'s `default_operating_point` block times - what a user of the reference runs (/root/reference src/prefs/mod.rs:203,
    src/ui/panels/md.rs:362-371, README.md:236-240): 64^3 rigid four-site OPC waters = 1,048,576 sites, SPME on the 200^3 mesh the
    library chooses, dt 2 fs, rc 10 A, skin 2 A - at ITS size: one wave per tile with the Ewald force table, merged dual-list body,
    one `water_step_kernel` pass per step, brick spread / 2-D transforms + own x pass / brick gather on the side stream.  20 steps from
    the random-orientation lattice (it heats to ~1150 K on the way: seven list rebuilds and nineteen pruning passes in those 20 steps),
    then the forces THE STEP LOOP LEFT BEHIND against the fp64 oracle's real-space sum + the numpy SPME of oracle/pme_ref.py on the same
    mesh + the erf corrections of the excluded pairs, the M sites' share spread onto their parents as the library does."""
    from molchanica_amd import _abi
    from oracle import pme_ref as P
    from tests.test_gpu_pme import excluded_pairs
    s = systems.opc_water_box(64, seed=5)
    beta = 0.3
    cfg = MdConfig(coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=beta, overrides=0, skin=2.0)        # bench.py default_operating_point_rate
    with mdx.MdState(s, cfg) as md:
        md.initialize_velocities(300.0, True, seed=1)
        md.step(0.002, None, 20)
        info, st = md.pair_launch_info(), md.stats()
        pos = md.positions()
        f_step = md.forces().astype(np.float64)            # NOT re-evaluated
        e = md.energy()                                     # plain list, energy flavour, the chain with its energy sums
        f_plain = md.forces().astype(np.float64)
    step = info["step"]
    assert (step["waves_per_tile"], step["dual"], step["half"], step["coulomb"], step["energy"]) == (1, 3, 1, 4, 0), info      # 4: Ewald real space from the force table
    assert step["tiles"] >= 12000 and info["water_step_launches"] >= 20 and info["water_step_mixed_launches"] == 0, info
    assert st["rebuild_count"] >= 3 and st["prune_passes"] >= 3 and st["rebuild_fallbacks"] == 0, (st["rebuild_count"], st["prune_passes"])
    x = pos.astype(np.float64)
    L = float(s.box_hi[0])
    box, q, K = np.full(3, L), s.charge.astype(np.float64), 200       # (198.6 A: the next 2-3-5-smooth size at ~1 A per mesh cell)
    cfg_real = MdConfig(coulomb_mode=_abi.COULOMB_EWALD, ewald_alpha=beta, overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED, skin=2.0)
    fo, eo = orc.forces(s, cfg_real, pos=x, use_cells=True)
    e_rec, f_rec = P.spme_recip(x, q, (0, 0, 0), box, beta, (K, K, K), 4)
    e_x, f_x = P.excluded_pair_correction(x, q, excluded_pairs(s), box, beta)
    e_rec += e_x + P.ewald_self_energy(q, beta) + P.ewald_background_energy(q, box, beta)
    f_rec += f_x
    vi, w = s.vsite_idx.astype(np.int64), s.vsite_w.astype(np.float64)          # site = O + w0 (H1 - O) + w1 (H2 - O): its force goes the same way
    fm = f_rec[vi[:, 0]].copy()
    f_rec[vi[:, 1]] += (1.0 - w[:, 0] - w[:, 1])[:, None] * fm
    f_rec[vi[:, 2]] += w[:, 0:1] * fm
    f_rec[vi[:, 3]] += w[:, 1:2] * fm
    f_rec[vi[:, 0]] = 0.0
    f_ref = fo + f_rec
    slack = orc.cutoff_slack(s, cfg_real, pos=pos, rel=4e-5)
    rec_rms = _rms(f_rec)
    # Bounds.  The real-space part carries SURVEY 8(c)'s per-atom bound (1e-4 max(|F|, 1) + the cutoff slack).  The mesh part is fp32 with
    # fixed-point LDS canvases against an fp64 numpy SPME: tests/test_gpu_pme.py bounds its rms at 2e-4 of the reciprocal force's rms on
    # small boxes - the same bound here; per atom 2e-3 of that rms on top.  Measured on the MI355X (tools/dbg/default_point_body.py):
    # rms 3.7e-4 kcal/mol/A = 1.1e-4 of the reciprocal rms (2e-5 of the total force's rms), worst atom 2.8e-3 = 0.4 of its bound.
    for name, f in (("step-loop", f_step), ("plain-list", f_plain)):
        assert np.abs(f[vi[:, 0]]).max() == 0.0, "a virtual site must not keep a force"
        err = np.linalg.norm(f - f_ref, axis=1)
        tol = 1e-4 * np.maximum(np.linalg.norm(f_ref, axis=1), 1.0) + slack + 2e-3 * rec_rms
        print(f"default operating point, {name} forces: rms {_rms(f - f_ref):.2e} = {_rms(f - f_ref) / rec_rms:.2e} of the reciprocal rms, worst atom {(err / tol).max():.2f} of its bound")
        assert _rms(f - f_ref) < 2e-4 * rec_rms, (name, _rms(f - f_ref) / rec_rms)
        assert (err <= tol).all(), (name, float((err / tol).max()))
    assert e["lj"] == pytest.approx(eo["lj"], rel=2e-6, abs=2e-2)
    assert e["coulomb"] == pytest.approx(eo["coulomb"], rel=5e-6)             # (measured 1.5e-6: fp32 pair terms, fp64 sums)
    assert e["coulomb_recip"] == pytest.approx(e_rec, rel=2e-5)               # (measured 2.7e-6; tests/test_gpu_pme.py's bound)
