"""Seeded synthetic inputs S0–S4 for the MD hot path.

The reference commits no structures or parameter files (/root/reference .gitignore:15-34), so
every configuration of BASELINE.json is generated here from a seed: flexible 3-site TIP3P water
(standard Amber values, not from the reference tree), GAFF-like ligands and bonded chains.
Pure numpy host code, setup-time only.
"""
from __future__ import annotations

import math

import numpy as np

from ._abi import MdSystem
from . import topology as topo

KB = 0.0019872041          # kcal/mol/K
ACC_CONV = 418.4           # kcal/mol/Å/Da -> Å/ps²

# flexible TIP3P
TIP3P = dict(o_sigma=3.15061, o_eps=0.1521, q_o=-0.834, q_h=0.417, r_oh=0.9572, k_b=553.0,
             theta=math.radians(104.52), k_theta=100.0, m_o=15.9994, m_h=1.008)

# GAFF-like LJ table for the synthetic solutes: (sigma Å, eps kcal/mol)
SOLUTE_TYPES = [(3.39967, 0.0860), (3.25000, 0.1700), (2.95992, 0.2100), (2.0, 0.0157)]
SOLUTE_MASS = [12.011, 14.007, 15.999, 1.008]


def maxwell_boltzmann(mass: np.ndarray, temp: float, rng: np.random.Generator) -> np.ndarray:
    sig = np.sqrt(KB * temp * ACC_CONV / mass)[:, None]
    v = rng.normal(size=(mass.size, 3)) * sig
    p = (v * mass[:, None]).sum(0) / mass.sum()
    return (v - p).astype(np.float32)


def _random_rotations(n: int, rng: np.random.Generator) -> np.ndarray:
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    r = np.empty((n, 3, 3))
    r[:, 0, 0] = 1 - 2 * (y * y + z * z); r[:, 0, 1] = 2 * (x * y - z * w); r[:, 0, 2] = 2 * (x * z + y * w)
    r[:, 1, 0] = 2 * (x * y + z * w); r[:, 1, 1] = 1 - 2 * (x * x + z * z); r[:, 1, 2] = 2 * (y * z - x * w)
    r[:, 2, 0] = 2 * (x * z - y * w); r[:, 2, 1] = 2 * (y * z + x * w); r[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return r


def _water_atoms(sites: np.ndarray, rng: np.random.Generator, jitter: float):
    """sites [W,3] -> pos [3W,3] (O,H,H per water), randomly oriented."""
    w = sites.shape[0]
    t = TIP3P
    h = np.array([[t["r_oh"] * math.sin(t["theta"] / 2), t["r_oh"] * math.cos(t["theta"] / 2), 0.0],
                  [-t["r_oh"] * math.sin(t["theta"] / 2), t["r_oh"] * math.cos(t["theta"] / 2), 0.0]])
    rot = _random_rotations(w, rng)
    pos = np.empty((w, 3, 3))
    pos[:, 0] = sites
    pos[:, 1] = sites + rot @ h[0]
    pos[:, 2] = sites + rot @ h[1]
    pos = pos.reshape(-1, 3)
    if jitter > 0:
        pos = pos + rng.normal(scale=jitter, size=pos.shape)
    return pos


def _water_topology(first_atom: int, w: int, type_o: int, type_h: int):
    t = TIP3P
    o = first_atom + 3 * np.arange(w, dtype=np.int64)
    bonds = np.stack([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1)], 1).reshape(-1, 2)
    angles = np.stack([o + 1, o, o + 2], 1)
    excl_pairs = np.concatenate([bonds, np.stack([o + 1, o + 2], 1)], 0)
    return dict(
        bonds=bonds, bond_k=np.full(2 * w, t["k_b"]), bond_r0=np.full(2 * w, t["r_oh"]),
        angles=angles, angle_k=np.full(w, t["k_theta"]), angle_t0=np.full(w, t["theta"]),
        excl_pairs=excl_pairs,
        mass=np.tile([t["m_o"], t["m_h"], t["m_h"]], w),
        charge=np.tile([t["q_o"], t["q_h"], t["q_h"]], w),
        lj_type=np.tile([type_o, type_h, type_h], w),
    )


# OPC 4-site water (Izadi, Anandakrishnan, Onufriev 2014), the reference's water model
# (README.md:239; md.water[i].{o,h0,h1,m}, src/properties/sol_shrinking_box.rs:605-613)
OPC = dict(o_sigma=3.16655, o_eps=0.21280, q_h=0.6791, q_m=-1.3582, r_oh=0.8724,
           theta=math.radians(103.6), r_om=0.1594, m_o=15.9994, m_h=1.008)


def opc_water_box(n_side: int = 6, seed: int = 5, spacing: float = 3.1034, temp: float = 300.0) -> MdSystem:
    """n_side³ rigid 4-site OPC waters: three distance constraints per water, the M site is a
    massless virtual site (flagged static) carrying the negative charge."""
    rng = np.random.default_rng(seed)
    w = n_side ** 3
    box = n_side * spacing
    g = (np.arange(n_side) + 0.5) * spacing
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    o = OPC
    hx, hy = o["r_oh"] * math.sin(o["theta"] / 2), o["r_oh"] * math.cos(o["theta"] / 2)
    a = o["r_om"] / (2.0 * hy)
    rot = _random_rotations(w, rng)
    local = np.array([[0, 0, 0], [hx, hy, 0], [-hx, hy, 0], [0, o["r_om"], 0]], dtype=np.float64)
    pos = (sites[:, None, :] + np.einsum("wij,kj->wki", rot, local)).reshape(-1, 3)
    base = 4 * np.arange(w, dtype=np.int64)
    cons = np.stack([np.stack([base, base + 1], 1), np.stack([base, base + 2], 1),
                     np.stack([base + 1, base + 2], 1)], 1).reshape(-1, 2)
    clen = np.tile([o["r_oh"], o["r_oh"], 2 * hx], w)
    pairs = np.concatenate([np.stack([base + i, base + j], 1) for i in range(4) for j in range(i + 1, 4)], 0)
    off, idx = topo.csr_from_pairs(4 * w, pairs)
    mass = np.tile([o["m_o"], o["m_h"], o["m_h"], 0.0], w).astype(np.float32)
    flags = np.tile([0, 0, 0, 1], w).astype(np.uint8)
    vel = np.zeros((4 * w, 3), dtype=np.float32)
    real = flags == 0
    vel[real] = maxwell_boltzmann(mass[real], temp, np.random.default_rng(seed + 100))
    return MdSystem(
        pos=pos, mass=mass, charge=np.tile([0.0, o["q_h"], o["q_h"], o["q_m"]], w),
        lj_type=np.tile([0, 1, 1, 1], w), lj_sigma=[o["o_sigma"], 0.0], lj_eps=[o["o_eps"], 0.0], vel=vel, flags=flags,
        excl_offsets=off, excl_idx=idx, mol_start=base, constraint_idx=cons, constraint_len=clen,
        vsite_idx=np.stack([base + 3, base, base + 1, base + 2], 1), vsite_w=np.tile([a, a], (w, 1)),
        periodic=True, box_lo=(0, 0, 0), box_hi=(box, box, box), name=f"opc{4 * w}",
    ).normalise()


def water_box(n_side: int = 6, seed: int = 5, spacing: float = 3.1034, jitter: float = 0.05,
              temp: float = 300.0, name: str | None = None, rigid: bool = False) -> MdSystem:
    """n_side³ flexible TIP3P waters on a jittered lattice.  n_side=70 is S4/C5 `water1M`
    (1,029,000 atoms, 217.24 Å cube).  rigid=True replaces the bond/angle terms by three distance
    constraints per water (jitter is then applied to whole molecules)."""
    if rigid:
        s = water_box(n_side, seed, spacing, 0.0, temp, name, rigid=False)
        t = TIP3P
        base = 3 * np.arange(n_side ** 3, dtype=np.int64)
        hh = 2 * t["r_oh"] * math.sin(t["theta"] / 2)
        s.constraint_idx = np.stack([np.stack([base, base + 1], 1), np.stack([base, base + 2], 1),
                                     np.stack([base + 1, base + 2], 1)], 1).reshape(-1, 2)
        s.constraint_len = np.tile([t["r_oh"], t["r_oh"], hh], n_side ** 3)
        s.bond_idx = np.zeros((0, 2), np.uint32); s.bond_k = np.zeros(0); s.bond_r0 = np.zeros(0)
        s.angle_idx = np.zeros((0, 3), np.uint32); s.angle_k = np.zeros(0); s.angle_theta0 = np.zeros(0)
        s.name = f"rigidwater{s.n_atoms}"
        return s.normalise()
    rng = np.random.default_rng(seed)
    w = n_side ** 3
    box = n_side * spacing
    g = (np.arange(n_side) + 0.5) * spacing
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    pos = _water_atoms(sites, rng, jitter)
    tp = _water_topology(0, w, 0, 1)
    off, idx = topo.csr_from_pairs(3 * w, tp["excl_pairs"])
    mass = tp["mass"].astype(np.float32)
    return MdSystem(
        pos=pos, mass=mass, charge=tp["charge"], lj_type=tp["lj_type"],
        lj_sigma=[TIP3P["o_sigma"], 0.0], lj_eps=[TIP3P["o_eps"], 0.0],
        vel=maxwell_boltzmann(mass, temp, np.random.default_rng(seed + 100)),
        bond_idx=tp["bonds"], bond_k=tp["bond_k"], bond_r0=tp["bond_r0"],
        angle_idx=tp["angles"], angle_k=tp["angle_k"], angle_theta0=tp["angle_t0"],
        excl_offsets=off, excl_idx=idx, mol_start=3 * np.arange(w),
        periodic=True, box_lo=(0, 0, 0), box_hi=(box, box, box),
        name=name or f"water{3 * w}",
    ).normalise()


# ---------------------------------------------------------------------------------------------
def _solute_from_geometry(pos: np.ndarray, bonds: np.ndarray, types: np.ndarray,
                          rng: np.random.Generator, charge_sigma: float = 0.3):
    """Parameterise a bonded solute around its generated geometry (equilibrium values are the
    actual geometry perturbed by 2 %, so every bonded term carries a non-zero force)."""
    n = pos.shape[0]
    adj = topo.adjacency(n, bonds)
    angles = topo.angles_from_bonds(adj)
    dihedrals = topo.dihedrals_from_bonds(adj)
    off, idx, p14 = topo.exclusions_and_pairs14(n, adj)
    b = np.asarray(bonds, dtype=np.int64)
    r = np.linalg.norm(pos[b[:, 0]] - pos[b[:, 1]], axis=1)
    bond_r0 = r * (1 + rng.normal(scale=0.02, size=r.size))
    bond_k = rng.uniform(250.0, 450.0, size=r.size)
    if angles.shape[0]:
        a = angles.astype(np.int64)
        v1 = pos[a[:, 0]] - pos[a[:, 1]]
        v2 = pos[a[:, 2]] - pos[a[:, 1]]
        cs = (v1 * v2).sum(1) / (np.linalg.norm(v1, axis=1) * np.linalg.norm(v2, axis=1))
        th = np.arccos(np.clip(cs, -1, 1))
        angle_t0 = np.clip(th * (1 + rng.normal(scale=0.02, size=th.size)), 0.3, math.pi - 0.05)
        angle_k = rng.uniform(40.0, 90.0, size=th.size)
    else:
        angle_t0 = angle_k = np.zeros(0)
    nd = dihedrals.shape[0]
    dih_v = rng.uniform(0.1, 2.0, size=nd)
    dih_n = rng.integers(1, 4, size=nd)
    dih_phase = rng.integers(0, 2, size=nd) * math.pi
    # H-like side atoms (type 3) carry a small positive charge like real hydrogens: a negative
    # one would pull TIP3P hydrogens (which have no LJ term) into a Coulomb singularity
    q = rng.normal(scale=charge_sigma, size=n)
    h_like = np.asarray(types) == 3
    q[h_like] = rng.uniform(0.03, 0.15, size=int(h_like.sum()))
    heavy = ~h_like
    if heavy.any():
        q[heavy] -= q.sum() / heavy.sum()
    else:
        q -= q.mean()
    return dict(angles=angles, dihedrals=dihedrals, excl_off=off, excl_idx=idx, p14=p14,
                bond_k=bond_k, bond_r0=bond_r0, angle_k=angle_k, angle_t0=angle_t0,
                dih_v=dih_v, dih_n=dih_n, dih_phase=dih_phase, charge=q,
                mass=np.asarray(SOLUTE_MASS)[types])


def lig50(seed: int = 1, n_atoms: int = 50) -> MdSystem:
    """S0/C1: random bonded tree (degree <= 4), vacuum, no cutoff — the editor's ~50-atom ligand
    (/root/reference src/mol_editor/mod.rs:826-912)."""
    rng = np.random.default_rng(seed)
    pos = np.zeros((n_atoms, 3))
    deg = np.zeros(n_atoms, dtype=int)
    bonds = []
    k = 1
    while k < n_atoms:
        parent = int(rng.integers(0, k))
        if deg[parent] >= (3 if parent else 4):
            continue
        for _ in range(200):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            cand = pos[parent] + 1.5 * d
            dist = np.linalg.norm(pos[:k] - cand, axis=1)
            dist[parent] = 9.0
            if dist.min() >= 2.1:
                break
        else:
            continue
        pos[k] = cand
        bonds.append((parent, k))
        deg[parent] += 1
        deg[k] += 1
        k += 1
    bonds = np.asarray(bonds)
    types = rng.integers(0, 4, size=n_atoms)
    sp = _solute_from_geometry(pos, bonds, types, rng)
    sig = rng.uniform(2.5, 3.4, size=n_atoms)
    eps = rng.uniform(0.015, 0.21, size=n_atoms)
    return MdSystem(
        pos=pos, mass=sp["mass"], charge=sp["charge"], lj_type=np.arange(n_atoms),
        lj_sigma=sig, lj_eps=eps, vel=maxwell_boltzmann(sp["mass"], 300.0, np.random.default_rng(seed + 100)),
        bond_idx=bonds, bond_k=sp["bond_k"], bond_r0=sp["bond_r0"],
        angle_idx=sp["angles"], angle_k=sp["angle_k"], angle_theta0=sp["angle_t0"],
        dihedral_idx=sp["dihedrals"], dihedral_v=sp["dih_v"], dihedral_phase=sp["dih_phase"],
        dihedral_n=sp["dih_n"], excl_offsets=sp["excl_off"], excl_idx=sp["excl_idx"],
        pairs14_idx=sp["p14"], mol_start=[0], periodic=False, name=f"lig{n_atoms}",
    ).normalise()


def _serpentine_chain(n_atoms: int, centre: np.ndarray, row_len: float, rng: np.random.Generator,
                      row_gap: float = 4.6, layer_gap: float = 5.2, rows_per_layer: int = 8):
    """Compact bonded chain: zig-zag backbone along a serpentine centre line (straight rows joined
    by semicircular U-turns, so no two non-bonded atoms come closer than ~3 Å), every second
    backbone atom carries one H-like side atom.  Returns pos, bonds, types."""
    step = 1.27
    n_back = int(math.ceil(n_atoms * 2 / 3))
    need = (n_back + 2) * step

    def arc(c, u, w, rad, n=12):   # half circle from c - rad*u to c + rad*u, bulging towards +w
        a = np.linspace(0.0, math.pi, n + 1)[1:]
        return [c - rad * math.cos(t) * u + rad * math.sin(t) * w for t in a]

    ex, ey, ez = np.eye(3)
    pts = [np.zeros(3)]
    total, row = 0.0, 0
    while total < need:
        layer, r = divmod(row, rows_per_layer)
        xdir = 1.0 if row % 2 == 0 else -1.0
        end = pts[-1] + xdir * row_len * ex
        pts.append(end)
        total += row_len
        last_in_layer = r == rows_per_layer - 1
        if last_in_layer:      # climb to the next layer: half circle in the x-z plane
            new = arc(end + 0.5 * layer_gap * ez, ez, xdir * ex, 0.5 * layer_gap)
            total += math.pi * 0.5 * layer_gap
        else:                  # next row of the same layer: half circle in the x-y plane
            ydir = 1.0 if layer % 2 == 0 else -1.0
            new = arc(end + 0.5 * row_gap * ydir * ey, ydir * ey, xdir * ex, 0.5 * row_gap)
            total += math.pi * 0.5 * row_gap
        pts.extend(new)
        row += 1
    pts = np.asarray(pts)
    seg = np.diff(pts, axis=0)
    seglen = np.linalg.norm(seg, axis=1)
    cum = np.concatenate([[0.0], np.cumsum(seglen)])
    s = (np.arange(n_back) + 0.5) * step
    k = np.clip(np.searchsorted(cum, s, side="right") - 1, 0, len(seg) - 1)
    back = pts[k] + seg[k] * ((s - cum[k]) / seglen[k])[:, None]
    tang = seg[k] / seglen[k][:, None]
    # zig-zag along the normal closest to z (y where the path itself runs along z)
    nrm = ez - (tang @ ez)[:, None] * tang
    bad = np.linalg.norm(nrm, axis=1) < 0.3
    nrm[bad] = ey - (tang[bad] @ ey)[:, None] * tang[bad]
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    zig = np.where(np.arange(n_back) % 2 == 0, 0.36, -0.36)
    back = back + zig[:, None] * nrm
    pos = [back]
    bonds = [np.stack([np.arange(n_back - 1), np.arange(1, n_back)], 1)]
    n_side = n_atoms - n_back
    owners = (np.arange(n_side) * 2) % n_back if n_side * 2 <= n_back else np.arange(n_side) % n_back
    side = back[owners] + (np.sign(zig[owners]) * 1.09)[:, None] * nrm[owners]
    pos.append(side)
    bonds.append(np.stack([owners, n_back + np.arange(n_side)], 1))
    pos = np.concatenate(pos) + rng.normal(scale=0.02, size=(n_atoms, 3))
    pos += centre - pos.mean(0)
    types = np.concatenate([rng.integers(0, 3, size=n_back), np.full(n_side, 3)])
    return pos, np.concatenate(bonds), types


def _helix(n_atoms: int, axis_origin: np.ndarray, phase: float, rng: np.random.Generator):
    """One helical strand (radius 9 Å, rise 1.0 Å per backbone atom) with side atoms pointing in.  The backbone
    zig-zags radially by ±0.4 Å so that its bond angles sit near 127°: on the smooth helix they were 177°, and a harmonic
    angle term whose minimum lies 3° from the straight angle has a cusp there (theta cannot pass 180°) - neither this engine
    nor the fp64 oracle conserves energy across it (+290 kcal/mol in 2000 steps for 6000 atoms, occasional blow-ups)."""
    n_back = int(math.ceil(n_atoms * 2 / 3))
    t = np.arange(n_back)
    ang = phase + t * 0.14
    rad = 9.0 + 0.4 * np.where(t % 2 == 0, 1.0, -1.0)
    back = np.stack([rad * np.cos(ang), rad * np.sin(ang), t * 0.34 * 3.0], 1)
    n_side = n_atoms - n_back
    owners = (np.arange(n_side) * 2) % n_back
    side = back[owners].copy()
    side[:, 0] *= (9.0 - 1.09) / 9.0
    side[:, 1] *= (9.0 - 1.09) / 9.0
    pos = np.concatenate([back, side]) + rng.normal(scale=0.03, size=(n_atoms, 3))
    pos += axis_origin
    bonds = np.concatenate([np.stack([t[:-1], t[1:]], 1), np.stack([owners, n_back + np.arange(n_side)], 1)])
    types = np.concatenate([rng.integers(0, 3, size=n_back), np.full(n_side, 3)])
    return pos, bonds, types


def _solvate(solutes: list, box: float, n_total: int | None, seed: int, name: str,
             clearance: float = 2.9) -> MdSystem:
    """Merge bonded solutes (pos, bonds, types, params) with lattice water up to n_total atoms."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed + 7)
    sol_pos = np.concatenate([s["pos"] for s in solutes]) if solutes else np.zeros((0, 3))
    n_sol = sol_pos.shape[0]
    tree = cKDTree(np.mod(sol_pos, box), boxsize=box) if n_sol else None
    n_side = int(round(box / 3.1034))
    while True:   # refine the water lattice until enough sites clear the solute
        sp = box / n_side
        g = (np.arange(n_side) + 0.5) * sp
        sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        if tree is not None:
            d, _ = tree.query(sites, k=1)
            sites = sites[d > clearance]
        if n_total is None or (n_total - n_sol) // 3 <= sites.shape[0]:
            break
        n_side += 1
        if sp < 2.7:
            raise ValueError(f"{name}: cannot place {(n_total - n_sol) // 3} waters")
    if n_total is not None:
        w = (n_total - n_sol) // 3
        keep = np.sort(rng.choice(sites.shape[0], size=w, replace=False))
        sites = sites[keep]
    w = sites.shape[0]
    wpos = _water_atoms(sites, rng, 0.05)
    n_types = len(SOLUTE_TYPES)
    tp = _water_topology(n_sol, w, n_types, n_types + 1)

    pos = np.concatenate([sol_pos, wpos])
    n = pos.shape[0]
    mass = np.concatenate([s["mass"] for s in solutes] + [tp["mass"]])
    charge = np.concatenate([s["charge"] for s in solutes] + [tp["charge"]])
    lj_type = np.concatenate([s["types"] for s in solutes] + [tp["lj_type"]])
    bonds, bk, br0, angles, ak, at0 = [], [], [], [], [], []
    dih, dv, dp, dn, p14, excl_pairs, mol_start = [], [], [], [], [], [], []
    base = 0
    for s in solutes:
        mol_start.append(base)
        bonds.append(s["bonds"] + base); bk.append(s["bond_k"]); br0.append(s["bond_r0"])
        angles.append(s["angles"].astype(np.int64) + base); ak.append(s["angle_k"]); at0.append(s["angle_t0"])
        dih.append(s["dihedrals"].astype(np.int64) + base); dv.append(s["dih_v"]); dp.append(s["dih_phase"]); dn.append(s["dih_n"])
        p14.append(s["p14"].astype(np.int64) + base)
        eo, ei = s["excl_off"].astype(np.int64), s["excl_idx"].astype(np.int64)
        ii = np.repeat(np.arange(s["pos"].shape[0]), np.diff(eo))
        m = ii < ei
        excl_pairs.append(np.stack([ii[m], ei[m]], 1) + base)
        base += s["pos"].shape[0]
    bonds.append(tp["bonds"]); bk.append(tp["bond_k"]); br0.append(tp["bond_r0"])
    angles.append(tp["angles"]); ak.append(tp["angle_k"]); at0.append(tp["angle_t0"])
    excl_pairs.append(tp["excl_pairs"])
    mol_start = np.concatenate([np.asarray(mol_start, dtype=np.int64), n_sol + 3 * np.arange(w)])
    off, idx = topo.csr_from_pairs(n, np.concatenate(excl_pairs))
    cat = lambda xs, shape: (np.concatenate(xs) if xs else np.zeros(shape))
    mass = mass.astype(np.float32)
    return MdSystem(
        pos=pos, mass=mass, charge=charge, lj_type=lj_type,
        lj_sigma=[t[0] for t in SOLUTE_TYPES] + [TIP3P["o_sigma"], 0.0],
        lj_eps=[t[1] for t in SOLUTE_TYPES] + [TIP3P["o_eps"], 0.0],
        vel=maxwell_boltzmann(mass, 300.0, np.random.default_rng(seed + 100)),
        bond_idx=np.concatenate(bonds), bond_k=np.concatenate(bk), bond_r0=np.concatenate(br0),
        angle_idx=np.concatenate(angles), angle_k=np.concatenate(ak), angle_theta0=np.concatenate(at0),
        dihedral_idx=cat(dih, (0, 4)), dihedral_v=cat(dv, 0), dihedral_phase=cat(dp, 0),
        dihedral_n=cat(dn, 0), excl_offsets=off, excl_idx=idx, pairs14_idx=cat(p14, (0, 2)),
        mol_start=mol_start, periodic=True, box_lo=(0, 0, 0), box_hi=(box, box, box), name=name,
    ).normalise()


def _solute_dict(pos, bonds, types, rng):
    sp = _solute_from_geometry(pos, bonds, types, rng)
    sp.update(pos=pos, bonds=np.asarray(bonds, dtype=np.int64), types=types)
    return sp


def dhfr23k(seed: int = 2, n_chain: int = 2489, n_total: int = 23558, box: float = 62.23) -> MdSystem:
    """S1/C2: DHFR/JAC-sized system — 2,489-atom bonded chain + 7,023 waters, 62.23 Å cube."""
    rng = np.random.default_rng(seed)
    pos, bonds, types = _serpentine_chain(n_chain, np.full(3, box / 2), 38.0, rng)
    return _solvate([_solute_dict(pos, bonds, types, rng)], box, n_total, seed, "dhfr23k")


def small_solvated(seed: int = 11, n_chain: int = 120, box: float = 26.0,
                   n_total: int | None = None) -> MdSystem:
    """A few-thousand-atom chain-in-water system for fast parity tests."""
    rng = np.random.default_rng(seed)
    pos, bonds, types = _serpentine_chain(n_chain, np.full(3, box / 2), 14.0, rng, rows_per_layer=3)
    return _solvate([_solute_dict(pos, bonds, types, rng)], box, n_total, seed, f"solv{n_chain}")


def complex50k(seed: int = 3, box: float = 80.0, n_total: int = 51200) -> MdSystem:
    """S2/C3: 5,000-atom chain + the 50-atom ligand placed ~8 Å from the chain centroid surface
    (cf. /root/reference src/docking/mod.rs:98-105) + water."""
    rng = np.random.default_rng(seed)
    pos, bonds, types = _serpentine_chain(5000, np.full(3, box / 2), 46.0, rng)
    chain = _solute_dict(pos, bonds, types, rng)
    lig = lig50(seed=1)
    lp = lig.pos.astype(np.float64)
    lp = lp - lp.mean(0)
    top = pos[:, 2].max()
    lp += np.array([box / 2, box / 2, top + 8.0 + (lp[:, 2].max() - lp[:, 2].min()) / 2])
    ltypes = rng.integers(0, 3, size=lp.shape[0])
    ligd = _solute_dict(lp, lig.bond_idx.astype(np.int64), ltypes, rng)
    return _solvate([chain, ligd], box, n_total, seed, "complex50k")


def small_complex(seed: int = 13, n_chain: int = 120, box: float = 30.0) -> MdSystem:
    """A few-thousand-atom receptor (chain) + the 50-atom ligand + water: complex50k's shape at test size (three molecule
    classes for `energy_potential_between_mols`: mol_start[0] the chain, [1] the ligand, [2:] the waters)."""
    rng = np.random.default_rng(seed)
    pos, bonds, types = _serpentine_chain(n_chain, np.full(3, box / 2), 14.0, rng, rows_per_layer=3)
    chain = _solute_dict(pos, bonds, types, rng)
    lig = lig50(seed=1)
    lp = lig.pos.astype(np.float64)
    lp = lp - lp.mean(0)
    lp += np.array([box / 2, box / 2, pos[:, 2].max() + 5.0 + (lp[:, 2].max() - lp[:, 2].min()) / 2])
    ligd = _solute_dict(lp, lig.bond_idx.astype(np.int64), rng.integers(0, 3, size=lp.shape[0]), rng)
    return _solvate([chain, ligd], box, None, seed, "small_complex")


def molecular_crystal(n_cells=(4, 2, 2), n_atoms: int = 12, spacing: float = 10.0, seed: int = 21) -> MdSystem:
    """n_cells[0] x [1] x [2] copies of one small flexible molecule on a lattice, randomly rotated, no solvent: the shape of the
    reference's crystal runs (/root/reference src/properties/crystal.rs:596-638; one `mol_start` entry per copy)."""
    rng = np.random.default_rng(seed)
    mol = lig50(seed=seed, n_atoms=n_atoms)
    p0 = mol.pos.astype(np.float64)
    p0 = p0 - p0.mean(0)
    rots = _random_rotations(int(np.prod(n_cells)), rng)
    solutes = []
    k = 0
    for ix in range(n_cells[0]):
        for iy in range(n_cells[1]):
            for iz in range(n_cells[2]):
                pos = p0 @ rots[k].T + (np.array([ix, iy, iz]) + 0.5) * spacing
                sd = _solute_dict(pos, mol.bond_idx.astype(np.int64), mol.lj_type.astype(np.int64) % 3, np.random.default_rng(seed + 1))
                solutes.append(sd)
                k += 1
    n = sum(sd["pos"].shape[0] for sd in solutes)
    bonds, bk, br0, angles, ak, at0, dih, dv, dp, dn, p14, excl_pairs, mol_start = ([] for _ in range(13))
    base = 0
    for sd in solutes:
        mol_start.append(base)
        bonds.append(sd["bonds"] + base); bk.append(sd["bond_k"]); br0.append(sd["bond_r0"])
        angles.append(sd["angles"].astype(np.int64) + base); ak.append(sd["angle_k"]); at0.append(sd["angle_t0"])
        dih.append(sd["dihedrals"].astype(np.int64) + base); dv.append(sd["dih_v"]); dp.append(sd["dih_phase"]); dn.append(sd["dih_n"])
        p14.append(sd["p14"].astype(np.int64) + base)
        eo, ei = sd["excl_off"].astype(np.int64), sd["excl_idx"].astype(np.int64)
        ii = np.repeat(np.arange(sd["pos"].shape[0]), np.diff(eo))
        m = ii < ei
        excl_pairs.append(np.stack([ii[m], ei[m]], 1) + base)
        base += sd["pos"].shape[0]
    off, idx = topo.csr_from_pairs(n, np.concatenate(excl_pairs))
    mass = np.concatenate([sd["mass"] for sd in solutes]).astype(np.float32)
    box = tuple(float(c * spacing) for c in n_cells)
    return MdSystem(
        pos=np.concatenate([sd["pos"] for sd in solutes]), mass=mass, charge=np.concatenate([sd["charge"] for sd in solutes]),
        lj_type=np.concatenate([sd["types"] for sd in solutes]), lj_sigma=[t[0] for t in SOLUTE_TYPES], lj_eps=[t[1] for t in SOLUTE_TYPES],
        vel=maxwell_boltzmann(mass, 300.0, np.random.default_rng(seed + 100)),
        bond_idx=np.concatenate(bonds), bond_k=np.concatenate(bk), bond_r0=np.concatenate(br0),
        angle_idx=np.concatenate(angles), angle_k=np.concatenate(ak), angle_theta0=np.concatenate(at0),
        dihedral_idx=np.concatenate(dih), dihedral_v=np.concatenate(dv), dihedral_phase=np.concatenate(dp), dihedral_n=np.concatenate(dn),
        excl_offsets=off, excl_idx=idx, pairs14_idx=np.concatenate(p14), mol_start=np.asarray(mol_start, dtype=np.uint32),
        periodic=True, box_lo=(0, 0, 0), box_hi=box, name="crystal%d" % len(solutes),
    ).normalise()


def dna100k(seed: int = 4, box: float = 100.0, n_total: int = 100000) -> MdSystem:
    """S3/C4: two ~800-atom helical strands + water, 100 Å cube."""
    rng = np.random.default_rng(seed)
    o = np.array([box / 2, box / 2, box / 2 - 0.5 * 534 * 0.34 * 3.0 * 0.5])
    a = _helix(800, o, 0.0, rng)
    b = _helix(800, o, math.pi, rng)
    return _solvate([_solute_dict(*a, rng), _solute_dict(*b, rng)], box, n_total - (n_total - 1600) % 3,
                    seed, "dna100k")


def water1m(seed: int = 5) -> MdSystem:
    """S4/C5: 70³ waters = 1,029,000 atoms, 217.24 Å cube."""
    return water_box(70, seed=seed, name="water1M")


BY_NAME = {
    "lig50": lig50, "dhfr23k": dhfr23k, "complex50k": complex50k, "dna100k": dna100k,
    "water1M": water1m,
}
