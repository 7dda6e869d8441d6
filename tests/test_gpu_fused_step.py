"""The large classes' step loop runs bonded gather + kick + drift as one pass over double-buffered positions
(mdx_integrate.hip: bonded_integrate_kernel; systems of >= 4096 tiles).  Every other GPU test uses systems below that size,
so this file covers the fused arrangement: against the separate launches (MDX_FUSE_BONDED_INTEGRATE=0, read per chunk)
and against the oracle (`MdState::step`, /root/reference src/md/mod.rs:716,748)."""
import os

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

pytestmark = pytest.mark.gpu


def _run(system, cfg, dt, bursts, fused):
    from molchanica_amd.md_state import MdState
    os.environ["MDX_FUSE_BONDED_INTEGRATE"] = "1" if fused else "0"
    try:
        with MdState(system, cfg) as md:
            for n in bursts:
                md.step(dt, None, n)
            return md.positions(), md.velocities(), md.forces(), md.stats(), md.energy()
    finally:
        os.environ.pop("MDX_FUSE_BONDED_INTEGRATE", None)


@pytest.fixture(scope="module")
def big_water():
    s = systems.water_box(45, seed=7)       # 273,375 atoms: 4.3 k tiles, the 4-waves-per-tile class
    assert s.n_atoms >= 262144
    return s


def _dev(a, b, box):
    d = a.astype(np.float64) - b
    d -= np.rint(d / box) * box
    return np.abs(d).max(), np.sqrt((d ** 2).sum(1).mean())


def test_fused_pass_follows_the_separate_launches(big_water):
    """70 steps from the hot synthetic lattice: list rebuilds, pruning passes and chunk boundaries (bursts of 1, 16, 5 and
    48 steps) all fall inside.  With the deterministic full-list pair kernel (nb_variant 2) the two arrangements differ only
    in the order f_pair + f_bonded is summed in: the trajectories stay together to fp32 rounding.  With the default half-list
    kernel (f32 atomics: the last bits vary run to run) the fused run must sit as close to an unfused run as two unfused runs
    sit to each other."""
    box = np.asarray(big_water.box_hi, np.float64) - np.asarray(big_water.box_lo, np.float64)
    bursts = (1, 16, 5, 48)
    # (reaction field: the force is continuous at the cutoff.  Under the shifted-potential default a pair that crosses the
    # cutoff one step earlier in one run kicks a hydrogen by ~1 kcal/mol/A for a step: 4e-3 A by the end, in either arrangement)
    det = MdConfig(skin=2.0, nb_variant=2, coulomb_mode=1)
    pa, va, fa, sa, ea = _run(big_water, det, 0.0005, bursts, fused=True)
    pb, vb, fb, sb, eb = _run(big_water, det, 0.0005, bursts, fused=False)
    assert sa["n_tiles"] >= 4096
    assert sa["rebuild_count"] >= 2 and sa["rebuild_count"] == sb["rebuild_count"]
    mx, rms = _dev(pa, pb, box)
    assert rms < 2e-5 and mx < 2e-3, (mx, rms)
    assert abs(ea["potential"] - eb["potential"]) < 2e-5 * abs(eb["potential"])
    assert abs(ea["kinetic"] - eb["kinetic"]) < 2e-5 * abs(eb["kinetic"])
    # the force array read back after the run is complete (pair + bonded): the last force call of a chunk is not deferred
    assert np.abs(fa - fb).max() < 0.05 + 1e-3 * np.abs(fb).max()

    cfg = MdConfig(skin=2.0)
    p1, *_ = _run(big_water, cfg, 0.0005, bursts, fused=True)
    p2, *_ = _run(big_water, cfg, 0.0005, bursts, fused=False)
    p3, *_ = _run(big_water, cfg, 0.0005, bursts, fused=False)
    mx12, rms12 = _dev(p1, p2, box)
    mx23, rms23 = _dev(p2, p3, box)
    assert rms12 < 4.0 * rms23 + 1e-5, (rms12, rms23, mx12, mx23)


def test_fused_pass_against_the_oracle(big_water):
    """18 steps (two chunks: the second starts with an unfused half kick) against the fp64 oracle."""
    from oracle import oracle
    cfg = MdConfig(skin=2.0)
    p, v, f, st, e = _run(big_water, cfg, 0.0005, (18,), fused=True)
    xo, vo, _ = oracle.step(big_water, cfg, 0.0005, 18, use_cells=True)
    box = np.asarray(big_water.box_hi, np.float64) - np.asarray(big_water.box_lo, np.float64)
    d = p.astype(np.float64) - xo
    d -= np.rint(d / box) * box
    assert np.sqrt((d ** 2).sum(1).mean()) < 2e-4, np.sqrt((d ** 2).sum(1).mean())
    assert np.abs(d).max() < 5e-3


def test_fused_pass_on_decomposed_handles(big_water):
    """Two ranks of the 273 k-atom box hold ~2.6 k tiles each - the class in which a decomposed handle's step loop runs the fused
    pass too (ghost slots are copied through; the halo unpack writes into the buffer the pass has just filled).  With the
    deterministic full-list kernel and reaction field the two ranks follow the single-device run, with the pass and with the
    separate launches (MDX_FUSE_BONDED_INTEGRATE_DD, read at library load: the separate arm is the per-chunk knob)."""
    import threading
    from molchanica_amd.md_state import Fabric, MdState
    box = np.asarray(big_water.box_hi, np.float64) - np.asarray(big_water.box_lo, np.float64)
    cfg = MdConfig(skin=2.0, nb_variant=2, coulomb_mode=1)
    with MdState(big_water, cfg) as md:
        md.step(0.0005, None, 40)
        ref_pos, ref_e = md.positions(), md.energy()

    def ranks(fused):
        os.environ["MDX_FUSE_BONDED_INTEGRATE"] = "1" if fused else "0"
        fabric = Fabric(2)
        res, errs = {}, []

        def run(rank):
            try:
                with MdState(big_water, cfg) as md:
                    md.comm_init_fabric(fabric, rank)
                    md.step(0.0005, None, 24)
                    md.profile(1)                      # (the per-kernel brackets count the fused launches)
                    md.step(0.0005, None, 16)
                    md.profile(0)
                    res[rank] = (md.positions(), md.energy(), md.stats())
            except BaseException as e:   # pragma: no cover
                errs.append(e)
                fabric.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        os.environ.pop("MDX_FUSE_BONDED_INTEGRATE", None)
        if errs:
            raise errs[0]
        return res

    for fused in (True, False):
        res = ranks(fused)
        for r in (0, 1):
            pos, e, st = res[r]
            assert st["n_tiles"] >= 2048                      # the rank is in the class that fuses
            assert (st["fused_launches"] > 0) == fused
            mx, rms = _dev(pos, ref_pos, box)
            assert rms < 5e-5 and mx < 5e-3, (fused, r, mx, rms)
            assert abs(e["potential"] - ref_e["potential"]) < 5e-5 * abs(ref_e["potential"])
