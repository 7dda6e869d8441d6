"""GPU parity of the SURVEY §8f rows against the oracle: minimiser, velocity initialisation,
thermostats, centre-of-mass drift removal, snapshots."""
import math

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

pytestmark = pytest.mark.gpu
NOCUT = dict(lj_cutoff=0.0, coulomb_cutoff=0.0)
KB = 0.0019872041


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1
    return md_state


def test_minimize_energy_matches_oracle(mdx, orc):
    """md.minimize_energy(dev, iters, None) (src/ui/mol_editor.rs:375)."""
    s = systems.lig50()
    cfg = MdConfig(**NOCUT)
    for n in (10, 60):
        with mdx.MdState(s, cfg) as md:
            v0 = md.velocities()
            e, it = md.minimize_energy(n)
            x = md.positions().astype(np.float64)
            assert np.array_equal(md.velocities(), v0), "the minimiser must leave velocities alone"
        xo, eo, ito = orc.minimize(s, cfg, n)
        assert it == ito == n
        assert e["potential"] == pytest.approx(eo["potential"], rel=2e-4, abs=2e-3)
        assert math.sqrt(((x - xo) ** 2).sum(1).mean()) < 2e-3
    # periodic, with neighbour rebuilds on the way and a force tolerance stop
    s = systems.small_solvated()
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.0, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        e0 = md.energy()["potential"]
        e, it = md.minimize_energy(40)
        x = md.positions().astype(np.float64)
    xo, eo, _ = orc.minimize(s, cfg, 40, use_cells=True)
    # 40 accept/reject decisions on a truncated-LJ surface: the f32 and f64 descents may take a
    # different branch once; both must have gained the same ~4.7e3 kcal/mol to within 0.2 %
    assert e["potential"] < e0 and e["potential"] == pytest.approx(eo["potential"], rel=2e-3)
    L = np.array(s.box_hi)
    d = x - xo
    d -= np.round(d / L) * L
    assert math.sqrt((d ** 2).sum(1).mean()) < 2e-2


def test_minimize_with_external_forces_and_static_atoms(mdx, orc):
    s = systems.lig50()
    s.flags = np.zeros(50, np.uint8)
    s.flags[:8] = 1
    ext = np.zeros((50, 3), np.float32)
    ext[20:30, 1] = 15.0
    cfg = MdConfig(**NOCUT)
    with mdx.MdState(s, cfg) as md:
        e, it = md.minimize_energy(25, ext)
        x = md.positions().astype(np.float64)
    xo, eo, _ = orc.minimize(s, cfg, 25, ext=ext)
    assert np.array_equal(x[:8], s.pos[:8].astype(np.float64))
    assert math.sqrt(((x - xo) ** 2).sum(1).mean()) < 2e-3


def test_minimizer_follows_an_external_pull_from_a_relaxed_structure(mdx, orc):
    """The acceptance energy includes the work of the external forces (src/mol_alignment.rs:356 hands the pulling
    forces to the minimiser): a relaxed ligand moves with the pull instead of refusing every step."""
    s = systems.lig50()
    cfg = MdConfig(**NOCUT)
    xr, _, _ = orc.minimize(s, cfg, 400, f_tol=0.5)          # relax first (oracle), then pull half of it
    s.pos = xr.astype(np.float32)
    ext = np.zeros((50, 3), np.float32)
    ext[25:, 0] = 4.0; ext[:25, 0] = -4.0
    with mdx.MdState(s, cfg) as md:
        e0 = md.energy()["potential"]
        e, it = md.minimize_energy(60, ext)
        x = md.positions().astype(np.float64)
    xo, eo, ito = orc.minimize(s, cfg, 60, ext=ext)
    work = float((ext * (x - s.pos)).sum())
    assert work > 0.5, "the molecule did not follow the pull"
    assert e["potential"] - work < e0, "U - F.x must have dropped"
    assert e["potential"] == pytest.approx(eo["potential"], rel=5e-3, abs=5e-2)
    assert math.sqrt(((x - xo) ** 2).sum(1).mean()) < 5e-3


def test_initialize_velocities_equals_oracle(mdx, orc):
    s = systems.water_box(6, seed=3)
    with mdx.MdState(s, MdConfig(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.5)) as md:
        md.initialize_velocities(310.0, True, seed=1234)
        v = md.velocities()
        t = md.energy()["temperature"]
    vo = orc.init_velocities(s, 310.0, True, 1234)
    assert np.array_equal(v, vo.astype(np.float32)), "same seed must mean the same velocities, bit for bit"
    assert t == pytest.approx(310.0, rel=0.1)


@pytest.mark.parametrize("kind", [1, 2])
def test_thermostat_matches_oracle(mdx, orc, kind):
    """VerletVelocity{thermostat: Some(tau)} with temp_target (src/ui/panels/md.rs:296-305)."""
    s = systems.water_box(6, seed=4, jitter=0.0)
    cfg = MdConfig(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.5, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        md.set_thermostat(kind, 280.0, 0.05, every_n_steps=10, seed=99)
        md.set_zero_com_drift(True)
        temps = []
        for _ in range(8):
            md.step(0.0005, None, 10)
            temps.append(md.energy()["temperature"])
        v = md.velocities().astype(np.float64)
    xo, vo, to = orc.step_thermo(s, cfg, 0.0005, 80, kind, 280.0, 0.05, 10, 99, zero_com=True)
    assert np.allclose(temps, to, rtol=2e-3), (temps, to)
    assert np.abs((v * s.mass[:, None]).sum(0)).max() < 0.5
    assert math.sqrt(((v - vo) ** 2).sum(1).mean()) < 0.05


def test_csvr_long_run_samples_target_temperature(mdx):
    s = systems.water_box(8, seed=5, jitter=0.0)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        md.set_thermostat(2, 300.0, 0.05, every_n_steps=10, seed=5)
        md.step(0.0005, None, 1500)
        ts = []
        for _ in range(40):
            md.step(0.0005, None, 25)
            ts.append(md.energy()["temperature"])
    assert np.mean(ts) == pytest.approx(300.0, rel=0.05), np.mean(ts)


def test_snapshots_cadence_and_contents(mdx):
    """snapshot_handlers.memory: Some(n) -> md.snapshots; flush_snapshot_queues (src/md/mod.rs:118-122)."""
    s = systems.water_box(6, seed=6)
    cfg = MdConfig(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.5, chunk_steps=16)
    with mdx.MdState(s, cfg) as md, mdx.MdState(s, cfg) as md2:
        md.set_snapshot_cadence(7, with_velocities=True)
        md.step(0.0005, None, 30)
        snaps = md.snapshots
        assert [sn["step"] for sn in snaps] == [7, 14, 21, 28]
        assert np.allclose([sn["time"] for sn in snaps], [0.0035, 0.007, 0.0105, 0.014], rtol=1e-5)
        assert md.time_ps == pytest.approx(0.015, rel=1e-5) and md.step_count == 30
        # the third snapshot equals the state of an independent run stopped at step 21
        md2.step(0.0005, None, 21)
        assert np.abs(snaps[2]["atom_posits"] - md2.positions()).max() < 1e-4
        e2 = md2.energy()
        assert snaps[2]["energy_data"]["potential"] == pytest.approx(e2["potential"], rel=1e-5, abs=1e-2)
        assert snaps[2]["atom_velocities"] is not None
        md.flush_snapshot_queues()
        assert md.snapshots == []
        md.set_snapshot_cadence(0)
        md.step(0.0005, None, 10)
        assert md.snapshots == []


def test_snapshot_handlers_each_with_its_own_cadence(mdx):
    """`snapshot_handlers {memory, dcd, gromacs: OutputControl{nstxout, nstvout, nstfout, ...}}` (src/properties/crystal.rs:335-342,
    src/ui/panels/md.rs:775-899): one snapshot per step that ANY handler names, tagged with the handlers that wanted it; velocities
    only where nstvout is due, forces only where nstfout is due - and those equal an independent run stopped there."""
    s = systems.water_box(6, seed=9)
    cfg = MdConfig(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.5, chunk_steps=16)
    with mdx.MdState(s, cfg) as md, mdx.MdState(s, cfg) as md2:
        md.set_snapshot_handlers(memory=4, dcd=10, nstvout=12, nstfout=20)
        md.step(0.0005, None, 41)
        snaps = md.snapshots
        assert [sn["step"] for sn in snaps] == [4, 8, 10, 12, 16, 20, 24, 28, 30, 32, 36, 40]
        by_step = {sn["step"]: sn for sn in snaps}
        assert by_step[8]["handlers"] == ["memory"] and by_step[10]["handlers"] == ["dcd"]
        assert by_step[12]["handlers"] == ["memory", "nstvout"] and by_step[20]["handlers"] == ["memory", "dcd", "nstfout"]
        assert by_step[40]["handlers"] == ["memory", "dcd", "nstfout"]
        for st, sn in by_step.items():
            assert (sn["atom_velocities"] is not None) == (st % 12 == 0), st
            assert ("atom_forces" in sn) == (st % 20 == 0), st
        md2.step(0.0005, None, 20)
        assert np.abs(by_step[20]["atom_posits"] - md2.positions()).max() < 1e-4
        f2 = md2.forces()
        assert np.abs(by_step[20]["atom_forces"] - f2).max() <= 2e-3 * max(1.0, float(np.abs(f2).max()))
        md2.step(0.0005, None, 4)
        # (flexible O-H: 1e-5 A of position difference is 2e-3 A/ps of hydrogen velocity one step later)
        assert np.abs(by_step[24]["atom_velocities"] - md2.velocities()).max() < 3e-2
        # all handlers off again; an unknown handler name is refused
        md.flush_snapshot_queues(); md.set_snapshot_handlers()
        md.step(0.0005, None, 12)
        assert md.snapshots == []
        with pytest.raises(Exception):
            md.set_snapshot_handlers(xtc=5)


def test_snapshots_follow_the_oracle_trajectory(mdx, orc):
    """Every stored snapshot - positions, velocities, per-term energies - against the ORACLE's trajectory stopped at the
    same step (not against a second engine run)."""
    s = systems.small_solvated()
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        md.set_snapshot_cadence(10, with_velocities=True)
        md.step(0.0005, None, 30)
        snaps = md.snapshots
    assert [sn["step"] for sn in snaps] == [10, 20, 30]
    L = np.array(s.box_hi) - np.array(s.box_lo)
    x, v = x0, v0
    for sn in snaps:
        x, v, _ = orc.step(s, cfg, 0.0005, 10, pos=x, vel=v)
        d = sn["atom_posits"].astype(np.float64) - x
        d -= np.round(d / L) * L
        assert math.sqrt((d ** 2).sum(1).mean()) < 5e-4, sn["step"]
        assert math.sqrt(((sn["atom_velocities"] - v) ** 2).sum(1).mean()) < 5e-2
        _, eo = orc.forces(s, cfg, pos=x)
        for k in ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14"):
            assert sn["energy_data"][k] == pytest.approx(eo[k], rel=2e-4, abs=0.5), (sn["step"], k)
        assert sn["energy_data"]["kinetic"] == pytest.approx(orc.kinetic(s, v), rel=1e-3)


def test_water_views_and_hydrogen_bonds_of_a_snapshot(mdx):
    """The reference keeps solvent water apart: `md.water[i].{o,h0,h1,m}.{posit,force}` (sol_shrinking_box.rs:605-613,
    780-786) and a Snapshot's `atom_posits` (non-water) + `water_o/h0/h1_posits` + `hydrogen_bonds`
    (src/md/viewer.rs:374-394, 917-960).  Views over the flat atom array, and the library's hydrogen-bond rule against
    a brute-force evaluation of the same rule."""
    s = systems.small_solvated()
    Lb = np.array(s.box_hi, dtype=np.float64)
    s.pos = np.mod(np.asarray(s.pos, dtype=np.float64) + 1.25, Lb).astype(np.float32)   # atom-wise wrapped, as a running box is: waters straddle the faces
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5)
    n_sol = int(s.mol_start[1])                       # one solute chain, then waters (O, H, H)
    n_w = (s.n_atoms - n_sol) // 3
    heavy = np.zeros(s.n_atoms, np.uint8)
    heavy[n_sol::3] = 1                               # water oxygens
    heavy[:n_sol:5] = 1                               # a few solute atoms play N/O
    with mdx.MdState(s, cfg) as md:
        md.set_water_layout(n_sol, n_w, 3)
        md.set_hbond_detection(heavy, 2.5, 120.0)
        md.set_snapshot_cadence(5)
        md.step(0.0005, None, 10)
        pos, frc = md.positions(), md.forces()
        w, wf = md.water("posit"), md.water("force")
        snaps = md.snapshots
    assert np.array_equal(w["o"], pos[n_sol::3])
    for key, off in (("h0", 1), ("h1", 2)):          # the same atoms, handed out WHOLE: in the image next to their oxygen
        dw = w[key].astype(np.float64) - pos[n_sol + off::3]
        assert np.abs(dw - np.round(dw / Lb) * Lb).max() < 1e-4
        assert np.linalg.norm(w[key] - w["o"], axis=1).max() < 1.3
    assert (np.abs(pos[n_sol + 1::3] - pos[n_sol::3]).max(axis=1) > 0.5 * Lb[0]).sum() > 5, "no water straddles a face in the flat array"
    assert np.array_equal(wf["h1"], frc[n_sol + 2::3])
    sn = snaps[-1]
    assert sn["step"] == 10 and sn["atom_posits"].shape == (n_sol, 3) and sn["water_o_posits"].shape == (n_w, 3)
    L = np.array(s.box_hi, dtype=np.float64)
    dh = sn["water_h0_posits"].astype(np.float64) - sn["all_posits"][n_sol + 1::3]
    assert np.abs(dh - np.round(dh / L) * L).max() < 1e-4      # the same atoms, handed out in the image next to their oxygen
    assert np.linalg.norm(sn["water_h0_posits"] - sn["water_o_posits"], axis=1).max() < 1.3
    # brute force of the documented rule on the snapshot's coordinates
    P = sn["all_posits"].astype(np.float64)
    L = np.array(s.box_hi, np.float64) - np.array(s.box_lo, np.float64)
    donor_of = {}
    for a, b in np.concatenate([s.bond_idx.reshape(-1, 2)]):
        for h_, d_ in ((a, b), (b, a)):
            if s.mass[h_] < 1.6 and s.mass[d_] >= 1.6 and heavy[d_]:
                donor_of[int(h_)] = int(d_)
    acc = np.nonzero(heavy)[0]
    want = set()
    for h_, d_ in donor_of.items():
        dh = P[d_] - P[h_]; dh -= np.round(dh / L) * L
        ha = P[acc] - P[h_]; ha -= np.round(ha / L) * L
        r = np.linalg.norm(ha, axis=1)
        cs = (ha @ dh) / (np.maximum(r, 1e-12) * np.linalg.norm(dh))
        ok = (r <= 2.5) & (r > 0) & (cs <= math.cos(math.radians(120.0))) & (acc != d_)
        for a_ in acc[ok]:
            want.add((d_, int(a_), h_))

    def atom_of(ref):
        t, i = ref
        return i if t == 0 else n_sol + 3 * i + (t - 1)
    got = {(atom_of(b["donor"]), atom_of(b["acceptor"]), atom_of(b["hydrogen"])) for b in sn["energy_data"]["hydrogen_bonds"]}
    assert len(want) > 50, "the test configuration has no hydrogen bonds to find"
    assert got == want
    assert all(0.0 <= b["strength"] <= 1.0 for b in sn["energy_data"]["hydrogen_bonds"])
    assert any(b["donor"][0] == 1 and b["hydrogen"][0] in (2, 3) for b in sn["energy_data"]["hydrogen_bonds"]), "water donors use the water types"


def test_shrink_cell_towards_matches_the_oracle(mdx, orc):
    """`md.shrink_cell_towards(dev, target, cfg)` + `md.step(dev, DT, None)` per iteration, as the packing loop of
    src/properties/sol_shrinking_box.rs:989-995 does: cell, coordinates and the trajectory follow the oracle's."""
    s = systems.water_box(9, seed=41)                                   # 2187 atoms, 27.9 A box
    cfg = MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=1)
    lo0, hi0 = np.asarray(s.box_lo, np.float64), np.asarray(s.box_hi, np.float64)
    c = 0.5 * (lo0 + hi0)
    tlo, thi = c - 13.0, c + 13.0                                       # 26 A target cube
    dt, shrink = 0.0005, 0.05
    with mdx.MdState(s, cfg) as md:
        x, v = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        lo, hi = lo0.copy(), hi0.copy()
        sys_o = s
        for it in range(12):
            shrank = md.shrink_cell_towards(tlo, thi, shrink)
            lo, hi, x, shrank_o = orc.shrink_cell_towards(lo, hi, tlo, thi, shrink, x)
            assert shrank == shrank_o == True
            md.step(dt, None, 1)
            import dataclasses
            sys_o = dataclasses.replace(s, box_lo=tuple(float(a) for a in lo), box_hi=tuple(float(a) for a in hi))
            x, v, _ = orc.step(sys_o, cfg, dt, 1, pos=x, vel=v)
        blo, bhi = md.cell()
        assert np.allclose(blo, lo, atol=1e-4) and np.allclose(bhi, hi, atol=1e-4)
        assert np.allclose(np.asarray(bhi) - np.asarray(blo), (hi0 - lo0) - 12 * shrink, atol=1e-3)
        L = np.asarray(hi, np.float64) - np.asarray(lo, np.float64)
        d = md.positions().astype(np.float64) - x
        d -= np.round(d / L) * L
        assert math.sqrt((d ** 2).sum(1).mean()) < 1e-3
        # at the target the call reports "did not shrink" and leaves the state alone
        for _ in range(40):
            md.shrink_cell_towards(tlo, thi, 0.5)
        assert md.shrink_cell_towards(tlo, thi, 0.5) is False
        blo, bhi = md.cell()
        assert np.allclose(np.asarray(bhi) - np.asarray(blo), 26.0, atol=1e-3)
        # an edge below 2 (cutoff + skin) is refused with the state untouched
        with pytest.raises(mdx.ParamError):
            md.shrink_cell_towards(c - 5.0, c + 5.0, 20.0)
        assert np.allclose(np.asarray(md.cell()[1]) - np.asarray(md.cell()[0]), 26.0, atol=1e-3)


def test_library_chosen_verlet_skin(mdx):
    """`MdConfig.skin = 0`: the library starts at 2 A (or what the box allows) and walks to the skin with the best measured step
    rate (mdx_step, skin_autotune).  The skin only decides which pairs are LISTED: the dynamics must be those of a fixed-skin
    run - same energy conservation, same temperature - whatever the tuner does, and the tuning must end."""
    s = systems.water_box(10, seed=3)              # 31 A box: room for skins up to ~4 A at rc 7
    base = dict(lj_cutoff=7.0, coulomb_cutoff=7.0, coulomb_mode=1)
    out = {}
    for name, skin in (("fixed", 1.5), ("auto", 0.0)):
        with mdx.MdState(s, MdConfig(skin=skin, **base)) as md:
            md.minimize_energy(50); md.initialize_velocities(300.0, True, seed=4)
            md.step(0.0005, None, 200)
            e0 = md.energy()
            md.step(0.0005, None, 9000)
            e1 = md.energy()
            out[name] = (e0, e1, md.skin(), md.stats()["rebuild_count"])
    (e0f, e1f, _, _), (e0a, e1a, (skin_a, tuning), rb_a) = out["fixed"], out["auto"]
    assert not tuning, "the skin tuning never finished"
    assert 0.75 <= skin_a <= 4.0 and abs(skin_a * 4 - round(skin_a * 4)) < 1e-4, skin_a      # a multiple of 0.25 A inside its range
    assert rb_a > 10
    tot = lambda e: e["potential"] + e["kinetic"]
    drift_f, drift_a = abs(tot(e1f) - tot(e0f)) / s.n_atoms, abs(tot(e1a) - tot(e0a)) / s.n_atoms
    assert drift_a < max(2.0 * drift_f, 2e-3), (drift_a, drift_f)
    assert abs(e1a["temperature"] - e1f["temperature"]) < 25.0
