"""GPU parity of constraints (SHAKE/RATTLE) and virtual sites against the oracle — the reference's
default operating point: dt = 2 fs, constrained hydrogens, 4-site OPC water
(/root/reference src/prefs/mod.rs:203; src/ui/panels/md.rs:362-371; sol_shrinking_box.rs:605-613)."""
import math
import os

import numpy as np
import pytest

from molchanica_amd import MdConfig, systems
from molchanica_amd import _abi

pytestmark = pytest.mark.gpu
KB = 0.0019872041


@pytest.fixture(scope="module")
def mdx():
    from molchanica_amd import md_state
    assert md_state.device_count() >= 1
    return md_state


def bond_errors(s, x):
    b = s.constraint_idx.astype(int)
    d = x[b[:, 0]] - x[b[:, 1]]
    L = np.array(s.box_hi, dtype=np.float64)
    d -= np.round(d / L) * L
    return np.abs(np.linalg.norm(d, axis=1) - s.constraint_len) / s.constraint_len


@pytest.mark.parametrize("model", ["tip3p_rigid", "opc", "opc_clusters_in_slot_order"])
def test_rigid_water_parity_at_2fs(mdx, orc, model, monkeypatch):
    if model == "opc_clusters_in_slot_order":      # the cluster table laid out in slot order (the library's own choice from 32 k clusters on)
        monkeypatch.setenv("MDX_CONS_SORT_MIN", "1")
    s = systems.water_box(8, seed=3, rigid=True) if model == "tip3p_rigid" else systems.opc_water_box(8, seed=3)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        f = md.forces().astype(np.float64)
        e = md.energy()
        pos = md.positions()
        fo, eo = orc.forces(s, cfg, pos=pos.astype(np.float64))
        err = np.linalg.norm(f - fo, axis=1)
        tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + orc.cutoff_slack(s, cfg, pos=pos)
        assert (err <= tol).all(), float((err / tol).max())
        if model == "opc":
            assert np.abs(f[3::4]).max() == 0.0, "a virtual site must not keep a force"
        for k in ("lj", "coulomb"):
            assert e[k] == pytest.approx(eo[k], rel=5e-6, abs=2e-2)   # a few cutoff-boundary pairs x 1e-3 kcal/mol: LJ is truncated, not shifted
        dof = 3 * int((s.mass > 0).sum()) - s.constraint_idx.shape[0] - 3
        assert e["temperature"] == pytest.approx(2 * e["kinetic"] / (dof * KB), rel=1e-9)
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        b = s.constraint_idx.astype(int)
        assert bond_errors(s, x0).max() < 3e-5
        md.step(0.002, None, 50)                       # the reference's default dt (src/prefs/mod.rs:203)
        if os.environ.get("MDX_WATER_STEP", "1") != "0":     # round 6: every step of a rigid-water box is ONE pass (water_step_kernel)
            assert md.pair_launch_info()["water_step_launches"] >= 50      # (enqueued launches: the ones gated off behind a stale list count too)
        x, v = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x).max() < 3e-5, "constraints drifted"
        d = x[b[:, 0]] - x[b[:, 1]]
        d -= np.round(d / s.box_hi[0]) * s.box_hi[0]
        assert np.abs((d * (v[b[:, 0]] - v[b[:, 1]])).sum(1)).max() < 5e-3, "velocity along a constrained bond"
        rebuilds = md.stats()["rebuild_count"]
    xo, vo, _ = orc.step(s, cfg, 0.002, 50, pos=x0, vel=v0, use_cells=True)
    L = np.array(s.box_hi)
    dd = x - xo
    dd -= np.round(dd / L) * L
    rms = math.sqrt((dd ** 2).sum(1).mean())
    assert rms < 2e-3, f"constrained trajectory deviates: {rms:.2e} Å"
    assert rebuilds >= 2


@pytest.mark.parametrize("model", ["tip3p_rigid", "opc"])
def test_rigid_waters_straddling_box_faces(mdx, orc, model):
    """The state a running box is always in: atoms are wrapped into the cell one by one, so a water near a face has
    its sites on BOTH sides of it (the lattice the generator starts from never has).  The pair list fixes one image
    shift per cluster pair at the rebuild; SHAKE moves atoms by corrections and a virtual site must be rebuilt in the
    periodic image it is stored in, not next to its parent - otherwise every straddling OPC water interacts through the
    wrong image and the box heats (found in round 2: +15 kcal/mol/ps per water at 23 k sites, invisible to the lattice
    start of the test above).  Forces, energies, a dt = 2 fs trajectory and energy conservation against the oracle,
    which takes the minimum image per atom pair."""
    s = systems.water_box(6, seed=7, rigid=True) if model == "tip3p_rigid" else systems.opc_water_box(6, seed=7)
    L = np.array(s.box_hi, dtype=np.float64)
    s.pos = np.mod(np.asarray(s.pos, dtype=np.float64) + 1.25, L).astype(np.float32)     # the last lattice layer now straddles every upper face
    nsite = 3 if model == "tip3p_rigid" else 4
    w = s.pos.reshape(-1, nsite, 3)
    assert (np.abs(w[:, 1:] - w[:, :1]).max(axis=(1, 2)) > 0.5 * L[0]).sum() > 30, "no water straddles a face"
    cfg = MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        f, e = md.forces().astype(np.float64), md.energy()
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        fo, eo = orc.forces(s, cfg, pos=x0)
        err = np.linalg.norm(f - fo, axis=1)
        tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + orc.cutoff_slack(s, cfg, pos=x0)
        assert (err <= tol).all(), float((err / tol).max())
        for k in ("lj", "coulomb"):
            assert e[k] == pytest.approx(eo[k], rel=5e-6, abs=2e-2)
        t0 = e["potential"] + e["kinetic"]
        md.step(0.002, None, 60)
        x = md.positions().astype(np.float64)
        e1 = md.energy()
        assert md.stats()["rebuild_count"] >= 3        # every rebuild wraps the atoms again
        assert bond_errors(s, x).max() < 3e-5
    xo, vo, _ = orc.step(s, cfg, 0.002, 60, pos=x0, vel=v0, use_cells=True)
    d = x - xo
    d -= np.round(d / L) * L
    rms = math.sqrt((d ** 2).sum(1).mean())
    assert rms < 2e-3, f"trajectory deviates from the oracle: {rms:.2e} A"
    assert abs(e1["potential"] + e1["kinetic"] - t0) < 0.02 * e["kinetic"], "energy not conserved over 60 steps of NVE"


def test_opc_box_conserves_energy_after_equilibration(mdx):
    """NVE at the reference's default operating point (rigid OPC, dt 2 fs) from an equilibrated, wrapped state: the total
    energy of 1000 waters may wander by a fraction of a percent of the kinetic energy over 300 steps (it rose by
    4500 kcal/mol in 200 steps with the virtual sites rebuilt in the wrong image)."""
    s = systems.opc_water_box(10, seed=5)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        md.minimize_energy(200)
        md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.02, 1)
        md.step(0.001, None, 1500)
        md.set_thermostat(0, 300.0, 0.02, 1)
        e0 = md.energy()
        md.step(0.002, None, 300)
        e1 = md.energy()
        assert md.stats()["rebuild_count"] >= 10
    drift = (e1["potential"] + e1["kinetic"]) - (e0["potential"] + e0["kinetic"])
    assert abs(drift) < 0.01 * e0["kinetic"], f"dE = {drift:.1f} kcal/mol (kinetic {e0['kinetic']:.0f})"
    assert abs(e1["temperature"] - e0["temperature"]) < 25.0


def test_one_pass_water_step_against_the_three_launches(tmp_path):
    """Boxes of rigid water step through ONE kernel per step (water_step_kernel, mdx_constraints.hip: site-force spread + kick + drift +
    SETTLE + site placement; the tests above hold it against the oracle).  MDX_WATER_STEP=0 keeps the three launches it replaces:
    same trajectories, energies and virial to rounding (the fused pass takes the old positions as stored instead of x' - dt v')."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    arms = {}
    for arm in ("0", "1"):
        env = {k: v for k, v in os.environ.items() if not k.startswith("MDX_")}
        env["MDX_WATER_STEP"] = arm
        out = tmp_path / f"arm{arm}.npz"
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "water_step_child.py"), str(out)], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "WATER-STEP-CHILD-OK" in p.stdout, (p.stdout + p.stderr)[-2000:]
        arms[arm] = np.load(out)
    a, b = arms["0"], arms["1"]
    for name in ("tip3p_rigid", "opc", "opc_straddling_spme"):
        d = a[name + "_pos"].astype(np.float64) - b[name + "_pos"].astype(np.float64)
        d -= np.round(d / 24.8272) * 24.8272
        # 60 steps at 2 fs from a 300 K start.  Two runs of the SAME arm already part - the half-list pair kernel's atomics land in a
        # different order every run and the trajectories amplify that: measured over repeated runs (tools/dbg/water_ab_noise.py),
        # same arm against itself AND arm against arm alike: positions 1.4e-4 ... 1.0e-3 A rms, velocities 1e-2 ... 7e-2 A/ps,
        # potential 0.002 ... 0.13 kcal/mol, kinetic energy 9e-7 ... 7e-5 relative, virial 0.05 ... 0.8.  The bounds sit above that
        # noise; what pins each arm is the fp64 oracle (2e-3 A over 50 steps in the tests above).
        assert math.sqrt((d ** 2).sum(1).mean()) < 5e-3, (name, math.sqrt((d ** 2).sum(1).mean()))
        dv = a[name + "_vel"].astype(np.float64) - b[name + "_vel"].astype(np.float64)
        assert math.sqrt((dv ** 2).sum(1).mean()) < 0.3, (name, math.sqrt((dv ** 2).sum(1).mean()))
        ea, eb = a[name + "_e"], b[name + "_e"]
        assert abs(ea[0] - eb[0]) < 2e-5 * abs(ea[0]) + 0.5 and abs(ea[1] - eb[1]) < 5e-4 * ea[1], (name, ea, eb)
        assert abs(ea[2] - eb[2]) < 2e-3 * abs(ea[2]) + 3.0, (name, "virial", ea[2], eb[2])
        assert a[name + "_rebuilds"][0] >= 3
        if name.startswith("opc"):
            assert np.abs(b[name + "_frc"][3::4]).max() == 0.0, "a virtual site must not keep a force"


@pytest.mark.parametrize("mode", [1, 2])
def test_clusters_by_interaction_kind_change_the_list_not_the_forces(mdx, mode, monkeypatch):
    """Four-site water: oxygens carry the Lennard-Jones well and no charge, hydrogens and M sites the charges.  The tile assignment
    (mdx_grid.hip kind_tile_order) then forms clusters per kind and the pruning pass drops the cluster pairs between the kinds -
    fewer listed pairs, the same trajectory, forces and energies; a system without such atoms forced through the same ordering
    (MDX_KIND_CLUSTERS=1) comes out the same too."""
    s = systems.opc_water_box(10, seed=13)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=2.0, coulomb_mode=mode, ewald_alpha=0.35,
                   overrides=_abi.OVR_LONG_RANGE_RECIP_DISABLED if mode == 2 else 0)
    out = {}
    for arm in ("0", "1", None):
        if arm is None: monkeypatch.delenv("MDX_KIND_CLUSTERS")          # the library's own choice for a system this small: off
        else: monkeypatch.setenv("MDX_KIND_CLUSTERS", arm)
        with mdx.MdState(s, cfg) as md:
            md.step(0.002, None, 40)                    # (lists of the unfused first build and of the fused rebuilds alike)
            st = md.stats()
            out[arm] = (md.forces().astype(np.float64), md.energy(), st["n_cluster_pairs"], md.positions().astype(np.float64), st["rebuild_count"])
    (f0, e0, n0, p0, r0), (f1, e1, n1, p1, r1) = out["0"], out["1"]
    assert n1 < 0.95 * n0 and abs(out[None][2] - n0) < 0.01 * n0, (n0, n1, out[None][2])
    assert r0 >= 2 and r1 >= 2 and np.abs(p1 - p0).max() < 5e-3
    # forces and energies of the two arms at ONE geometry (a new set of positions is a rebuild through the fused chain)
    at = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("MDX_KIND_CLUSTERS", arm)
        with mdx.MdState(s, cfg) as md:
            md.set_positions(p1.astype(np.float32))
            at[arm] = (md.forces().astype(np.float64), md.energy(), md.stats()["n_cluster_pairs"])
    assert at["1"][2] < 0.95 * at["0"][2]
    scale = np.maximum(np.abs(at["0"][0]).max(1), 1.0)
    assert (np.abs(at["1"][0] - at["0"][0]).max(1) / scale).max() < 1e-4
    for k in ("lj", "coulomb", "potential"):
        assert at["1"][1][k] == pytest.approx(at["0"][1][k], rel=1e-5, abs=1e-3), k
    w = systems.small_solvated()
    wcfg = MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0, coulomb_mode=1)
    outw = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("MDX_KIND_CLUSTERS", arm)
        with mdx.MdState(w, wcfg) as md:
            md.step(0.0005, None, 30)
            outw[arm] = (md.positions().astype(np.float64), md.stats()["rebuild_count"])
    assert outw["1"][1] >= 1 and np.abs(outw["1"][0] - outw["0"][0]).max() < 2e-4


def test_xh_constraints_on_a_solvated_chain(mdx, orc):
    """HydrogenConstraint::Shake on the solute's X-H bonds; input geometry is off the constraint
    lengths by ~2 %, so creation has to project it first."""
    s = systems.small_solvated()
    h_side = np.nonzero(s.lj_type == 3)[0]
    is_h = np.zeros(s.n_atoms, bool)
    is_h[h_side] = True
    sel = is_h[s.bond_idx[:, 0]] | is_h[s.bond_idx[:, 1]]
    s.constraint_idx = s.bond_idx[sel].copy()
    s.constraint_len = s.bond_r0[sel].copy()
    assert sel.sum() > 20
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        md.forces()
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x0).max() < 3e-5
        md.step(0.001, None, 40)
        x = md.positions().astype(np.float64)
        assert bond_errors(s, x).max() < 3e-5
    xo, vo, _ = orc.step(s, cfg, 0.001, 40, pos=x0, vel=v0, use_cells=True)
    L = np.array(s.box_hi)
    d = x - xo
    d -= np.round(d / L) * L
    assert math.sqrt((d ** 2).sum(1).mean()) < 2e-3


def _methyl_box(n_side=3, spacing=4.6, seed=9):
    """n_side^3 methane-like molecules, three C-H bonds of each constrained (a 4-atom / 3-constraint cluster, the shape
    of a methyl group), the fourth flexible; angles flexible."""
    from molchanica_amd._abi import MdSystem
    from molchanica_amd import topology as topo
    rng = np.random.default_rng(seed)
    t = 1.0 / math.sqrt(3.0)
    tet = 1.09 * np.array([[t, t, t], [t, -t, -t], [-t, t, -t], [-t, -t, t]])
    g = (np.arange(n_side) + 0.5) * spacing
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    m = sites.shape[0]
    rot = systems._random_rotations(m, rng)
    pos = np.empty((m, 5, 3))
    pos[:, 0] = sites
    pos[:, 1:] = sites[:, None, :] + np.einsum("mij,kj->mki", rot, tet)
    pos = pos.reshape(-1, 3)
    c = 5 * np.arange(m)
    bonds = np.stack([np.repeat(c, 4), (c[:, None] + np.arange(1, 5)).ravel()], 1)
    angles = np.array([[c0 + a, c0, c0 + b] for c0 in c for a in range(1, 5) for b in range(a + 1, 5)])
    pairs = np.concatenate([bonds, angles[:, [0, 2]]], 0)
    off, idx = topo.csr_from_pairs(5 * m, pairs)
    mass = np.tile([12.011, 1.008, 1.008, 1.008, 1.008], m).astype(np.float32)
    cons = bonds.reshape(m, 4, 2)[:, :3].reshape(-1, 2)        # three of the four C-H bonds
    box = n_side * spacing
    return MdSystem(
        pos=pos, mass=mass, charge=np.tile([-0.4, 0.1, 0.1, 0.1, 0.1], m), lj_type=np.tile([0, 1, 1, 1, 1], m),
        lj_sigma=[3.39967, 2.0], lj_eps=[0.0860, 0.0157], vel=systems.maxwell_boltzmann(mass, 300.0, np.random.default_rng(seed + 1)),
        bond_idx=bonds, bond_k=np.full(len(bonds), 340.0), bond_r0=np.full(len(bonds), 1.09),
        angle_idx=angles, angle_k=np.full(len(angles), 35.0), angle_theta0=np.full(len(angles), math.radians(109.47)),
        excl_offsets=off, excl_idx=idx, mol_start=c, constraint_idx=cons, constraint_len=np.full(len(cons), 1.09),
        periodic=True, box_lo=(0, 0, 0), box_hi=(box, box, box), name="methyl",
    ).normalise()


def test_methyl_clusters_meet_the_tolerance_linear_maps_to(mdx, orc):
    """`HydrogenConstraint::Linear{order, iter}` (LINCS; /root/reference src/ui/panels/md.rs:363-366, the UI default) is served by
    the converged SHAKE / RATTLE solver (include/mdx.h, "constraints"): `order` and `iter` bound LINCS's truncation error, and
    the solver here iterates until every constraint of a cluster is within `constraint_tol` (1e-5 relative, tighter than what
    LINCS order 4 / iter 1 leaves), so there is nothing for them to select.  A methyl-shaped cluster - one heavy atom, three
    constrained hydrogens, the coupled case LINCS's matrix expansion exists for - must meet that tolerance at dt = 2 fs and
    follow the oracle's trajectory."""
    s = _methyl_box()
    cfg = MdConfig(lj_cutoff=5.0, coulomb_cutoff=5.0, skin=1.0, coulomb_mode=1)
    assert cfg.constraint_tol == pytest.approx(1e-5)
    with mdx.MdState(s, cfg) as md:
        md.forces()
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x0).max() < 3e-5
        md.step(0.002, None, 50)
        x, v = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x).max() < 3e-5, "a methyl cluster left the tolerance"
        b = s.constraint_idx.astype(int)
        d = x[b[:, 0]] - x[b[:, 1]]
        d -= np.round(d / s.box_hi[0]) * s.box_hi[0]
        assert np.abs((d * (v[b[:, 0]] - v[b[:, 1]])).sum(1)).max() < 5e-3, "velocity along a constrained bond"
    xo, vo, _ = orc.step(s, cfg, 0.002, 50, pos=x0, vel=v0, use_cells=True)
    L = np.array(s.box_hi)
    dd = x - xo
    dd -= np.round(dd / L) * L
    assert math.sqrt((dd ** 2).sum(1).mean()) < 2e-3


def test_hydrogen_constraint_kinds_are_mapped_not_ignored(mdx):
    """`HydrogenConstraint::{Shake{shake_tolerance}, Linear{order, iter}, Flexible}` as the UI hands it over
    (/root/reference src/ui/panels/md.rs:362-371): Shake sets the tolerance, Linear is MAPPED onto the converged solver with a
    tolerance no looser than LINCS would leave and says so, Flexible is refused on a system that was built with constraints."""
    s = _methyl_box()
    cfg = MdConfig(lj_cutoff=5.0, coulomb_cutoff=5.0, skin=1.0, coulomb_mode=1, constraint_tol=1e-3)
    with mdx.MdState(s, cfg) as md:
        text = md.set_hydrogen_constraint("linear", order=4, iters=1)
        assert "Linear{order 4, iter 1}" in text and "1.0e-04" in text, text
        md.forces(); md.step(0.002, None, 30)
        assert bond_errors(s, md.positions().astype(np.float64)).max() < 3e-4
        text = md.set_hydrogen_constraint("shake", shake_tolerance=1e-6)
        assert "1.0e-06" in text
        md.step(0.002, None, 10)
        assert bond_errors(s, md.positions().astype(np.float64)).max() < 5e-6
        with pytest.raises(mdx.ParamError, match="Flexible"):
            md.set_hydrogen_constraint("flexible")


def test_oversized_constraint_cluster_is_rejected(mdx):
    s = systems.lig50()
    s.constraint_idx = s.bond_idx[:12].copy()       # a connected chain of > 4 atoms
    s.constraint_len = s.bond_r0[:12].copy()
    with pytest.raises(mdx.ParamError, match="cluster"):
        mdx.MdState(s, MdConfig(lj_cutoff=0, coulomb_cutoff=0))


def _ammonium_in_water(n_side=12, n_ions=10, spacing=3.1034, seed=12):
    """n_ions NH4+ (all four N-H bonds constrained: a cluster of five atoms, the X-H4 star) among rigid TIP3P waters
    (three-atom clusters solved in closed form): `HydrogenConstraint` constrains "the hydrogens", whatever the heavy atom
    carries (/root/reference src/ui/panels/md.rs:362-371)."""
    from molchanica_amd._abi import MdSystem
    from molchanica_amd import topology as topo
    rng = np.random.default_rng(seed)
    g = (np.arange(n_side) + 0.5) * spacing
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    pick = np.sort(rng.choice(len(sites), size=n_ions, replace=False))
    ion_sites = sites[pick]
    wat_sites = np.delete(sites, pick, axis=0)
    w = len(wat_sites)
    t = 1.0 / math.sqrt(3.0)
    tet = 1.01 * np.array([[t, t, t], [t, -t, -t], [-t, t, -t], [-t, -t, t]])
    rot = systems._random_rotations(n_ions, rng)
    ip = np.empty((n_ions, 5, 3))
    ip[:, 0] = ion_sites
    ip[:, 1:] = ion_sites[:, None, :] + np.einsum("mij,kj->mki", rot, tet)
    wp = systems._water_atoms(wat_sites, rng, 0.0)
    pos = np.concatenate([ip.reshape(-1, 3), wp])
    n_i = 5 * n_ions
    c = 5 * np.arange(n_ions)
    nh = np.stack([np.repeat(c, 4), (c[:, None] + np.arange(1, 5)).ravel()], 1)
    angles = np.array([[c0 + a, c0, c0 + b] for c0 in c for a in range(1, 5) for b in range(a + 1, 5)])
    o = n_i + 3 * np.arange(w)
    tp = systems.TIP3P
    hh = 2 * tp["r_oh"] * math.sin(tp["theta"] / 2)
    wcons = np.stack([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1), np.stack([o + 1, o + 2], 1)], 1).reshape(-1, 2)
    cons = np.concatenate([nh, wcons])
    clen = np.concatenate([np.full(len(nh), 1.01), np.tile([tp["r_oh"], tp["r_oh"], hh], w)])
    pairs = np.concatenate([nh, angles[:, [0, 2]], wcons])
    off, idx = topo.csr_from_pairs(len(pos), pairs)
    mass = np.concatenate([np.tile([14.007, 1.008, 1.008, 1.008, 1.008], n_ions), np.tile([tp["m_o"], tp["m_h"], tp["m_h"]], w)]).astype(np.float32)
    box = n_side * spacing
    return MdSystem(
        pos=pos, mass=mass, charge=np.concatenate([np.tile([-0.4, 0.35, 0.35, 0.35, 0.35], n_ions), np.tile([tp["q_o"], tp["q_h"], tp["q_h"]], w)]),
        lj_type=np.concatenate([np.tile([2, 1, 1, 1, 1], n_ions), np.tile([0, 1, 1], w)]),
        lj_sigma=[tp["o_sigma"], 0.0, 3.25], lj_eps=[tp["o_eps"], 0.0, 0.17],
        vel=systems.maxwell_boltzmann(mass, 300.0, np.random.default_rng(seed + 1)),
        angle_idx=angles, angle_k=np.full(len(angles), 40.0), angle_theta0=np.full(len(angles), math.radians(109.47)),
        excl_offsets=off, excl_idx=idx, mol_start=np.concatenate([c, o]), constraint_idx=cons, constraint_len=clen,
        periodic=True, box_lo=(0, 0, 0), box_hi=(box, box, box), name="ammonium_in_water",
    ).normalise()


def test_ammonium_in_water_five_atom_clusters_at_2fs(mdx, orc):
    """X-H4 clusters (five atoms, four bonds from one centre) beside rigid waters at dt = 2 fs: SHAKE keeps every bond within the
    tolerance, RATTLE leaves no velocity along a bond, the trajectory follows the oracle's (generic iterative SHAKE / RATTLE in
    fp64) and a run on two ranks - a star is owned by the rank of its nitrogen - follows the single-GPU one."""
    from tests.test_gpu_comm import run_ranks, rms_dev
    s = _ammonium_in_water()
    cfg = MdConfig(lj_cutoff=6.5, coulomb_cutoff=6.5, skin=1.0, coulomb_mode=1, chunk_steps=8)
    L = np.array(s.box_hi)
    with mdx.MdState(s, cfg) as md:
        assert "clusters" in md.set_hydrogen_constraint("shake")
        md.minimize_energy(60)
        md.initialize_velocities(300.0, True, seed=4)
        md.forces()
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x0).max() < 3e-5
        md.step(0.002, None, 30)
        x, v = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x).max() < 3e-5, "a constrained bond left the tolerance"
        b = s.constraint_idx.astype(int)
        d = x[b[:, 0]] - x[b[:, 1]]
        d -= np.round(d / L) * L
        assert np.abs((d * (v[b[:, 0]] - v[b[:, 1]])).sum(1)).max() < 5e-3, "velocity along a constrained bond"
        e = md.energy()
        assert np.isfinite(e["pressure"]) and np.isfinite(e["virial"])
    xo, vo, _ = orc.step(s, cfg, 0.002, 30, pos=x0, vel=v0, use_cells=True)
    dd = x - xo
    dd -= np.round(dd / L) * L
    ions = slice(0, 5 * 10)
    assert math.sqrt((dd ** 2).sum(1).mean()) < 2e-3
    assert math.sqrt((dd[ions] ** 2).sum(1).mean()) < 2e-3, "the ammonium ions leave the oracle's trajectory"
    import dataclasses
    s2 = dataclasses.replace(s, pos=x0.astype(np.float32), vel=v0.astype(np.float32))
    with mdx.MdState(s2, cfg) as md:
        md.step(0.002, None, 20)
        p1 = md.positions()
    res = run_ranks(s2, cfg, 2, 20, dt=0.002)
    assert rms_dev(res[0]["pos"], p1, L) < 5e-4
    assert bond_errors(s, res[0]["pos"].astype(np.float64)).max() < 3e-5


def test_five_atom_cluster_that_is_not_a_star_is_rejected(mdx):
    s = systems.lig50()
    s.constraint_idx = s.bond_idx[:4].copy()        # the first four bonds of the generator's tree
    s.constraint_len = s.bond_r0[:4].copy()
    import collections
    deg = collections.Counter(s.constraint_idx.ravel().tolist())
    if max(deg.values()) == 4 and len(deg) == 5:
        pytest.skip("the generator's first four bonds happen to form a star")
    with pytest.raises(mdx.ParamError, match="cluster"):
        mdx.MdState(s, MdConfig(lj_cutoff=0, coulomb_cutoff=0))


def _chain_in_rigid_water(flexible=0):
    """A flexible chain with its X-H bonds constrained (HydrogenConstraint::Shake) in RIGID three-site water - what the reference's
    users run (solute + rigid water, /root/reference src/ui/panels/md.rs:362-371); the first `flexible` waters keep their bonds and
    angles instead (mobile atoms that belong to no cluster, like any unconstrained part of a solute)."""
    s = systems.small_solvated(n_chain=160, box=30.0)
    n_sol = int(s.mol_start[1])
    n_keep = n_sol + 3 * flexible                                            # atoms below this index keep their bonded terms
    t = systems.TIP3P
    nw = (s.n_atoms - n_sol) // 3
    keep_b = (s.bond_idx < n_keep).all(1)
    keep_a = (s.angle_idx < n_keep).all(1)
    h_side = np.nonzero(s.lj_type == 3)[0]
    is_h = np.zeros(s.n_atoms, bool); is_h[h_side] = True
    xh = (s.bond_idx < n_sol).all(1) & (is_h[s.bond_idx[:, 0]] | is_h[s.bond_idx[:, 1]])
    base = n_sol + 3 * np.arange(flexible, nw, dtype=np.int64)
    hh = 2 * t["r_oh"] * math.sin(t["theta"] / 2)
    wc = np.stack([np.stack([base, base + 1], 1), np.stack([base, base + 2], 1), np.stack([base + 1, base + 2], 1)], 1).reshape(-1, 2)
    s.constraint_idx = np.concatenate([s.bond_idx[xh].astype(np.int64), wc]).astype(np.uint32)
    s.constraint_len = np.concatenate([s.bond_r0[xh], np.tile([t["r_oh"], t["r_oh"], hh], len(base))]).astype(np.float32)
    keep_b &= ~xh
    s.bond_idx, s.bond_k, s.bond_r0 = s.bond_idx[keep_b], s.bond_k[keep_b], s.bond_r0[keep_b]
    s.angle_idx, s.angle_k, s.angle_theta0 = s.angle_idx[keep_a], s.angle_k[keep_a], s.angle_theta0[keep_a]
    return s, n_sol


@pytest.mark.parametrize("flexible", [0, 12])
def test_solute_in_rigid_water_one_pass_for_the_waters(mdx, orc, flexible):
    """Mixed system: the rigid waters go through water_step_kernel, the solute (and free atoms) through integrate_kernel - which skips
    the waters' slots - and its X-H clusters through the general solver - which skips the waters' records.  Against the oracle."""
    s, n_sol = _chain_in_rigid_water(flexible)
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    with mdx.MdState(s, cfg) as md:
        md.forces()
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        assert bond_errors(s, x0).max() < 3e-5
        for burst in (1, 9, 16, 14):
            md.step(0.001, None, burst)
        x = md.positions().astype(np.float64)
        assert bond_errors(s, x).max() < 3e-5
        f = md.forces().astype(np.float64)                      # what the step loop left behind: complete
        fo, _ = orc.forces(s, cfg, pos=x, use_cells=True)
        err = np.linalg.norm(f - fo, axis=1)
        tol = 1e-4 * np.maximum(np.linalg.norm(fo, axis=1), 1.0) + orc.cutoff_slack(s, cfg, pos=md.positions())
        assert (err <= tol).all(), float((err / tol).max())
        assert md.stats()["rebuild_count"] >= 2
        info = md.pair_launch_info()
        if os.environ.get("MDX_WATER_STEP_MIXED", "1") != "0" and os.environ.get("MDX_WATER_STEP", "1") != "0":
            assert info["water_step_mixed_launches"] >= 40, info
    xo, vo, _ = orc.step(s, cfg, 0.001, 40, pos=x0, vel=v0, use_cells=True)
    L = np.array(s.box_hi)
    d = x - xo
    d -= np.round(d / L) * L
    assert math.sqrt((d ** 2).sum(1).mean()) < 2e-3
