"""Host-side logic: topology derivation, seeded system generators, ABI marshalling. CPU only."""
import numpy as np
import pytest

from molchanica_amd import MdConfig, systems, topology as topo


def test_topology_butane_like():
    adj = topo.adjacency(5, [[0, 1], [1, 2], [2, 3], [2, 4]])
    ang = topo.angles_from_bonds(adj)
    assert sorted(map(tuple, ang.tolist())) == [(0, 1, 2), (1, 2, 3), (1, 2, 4), (3, 2, 4)]
    dih = topo.dihedrals_from_bonds(adj)
    assert sorted(map(tuple, dih.tolist())) == [(0, 1, 2, 3), (0, 1, 2, 4)]
    off, idx, p14 = topo.exclusions_and_pairs14(5, adj)
    assert idx[off[0]:off[1]].tolist() == [1, 2]
    assert idx[off[2]:off[3]].tolist() == [0, 1, 3, 4]
    assert sorted(map(tuple, p14.tolist())) == [(0, 3), (0, 4)]
    # symmetric CSR
    for i in range(5):
        for j in idx[off[i]:off[i + 1]]:
            assert i in idx[off[j]:off[j + 1]]


def test_ring_has_no_14_inside_13():
    # cyclobutane: every 1-4 candidate is also a 1-2 -> no 1-4 pairs
    adj = topo.adjacency(4, [[0, 1], [1, 2], [2, 3], [3, 0]])
    _, _, p14 = topo.exclusions_and_pairs14(4, adj)
    assert p14.shape[0] == 0


def test_csr_from_pairs_matches_loop():
    rng = np.random.default_rng(0)
    pairs = np.unique(np.sort(rng.integers(0, 50, size=(200, 2)), axis=1), axis=0)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    off, idx = topo.csr_from_pairs(50, pairs)
    ref = [set() for _ in range(50)]
    for a, b in pairs:
        ref[a].add(b)
        ref[b].add(a)
    for i in range(50):
        assert idx[off[i]:off[i + 1]].tolist() == sorted(ref[i])


def test_lig50_is_seeded_and_connected():
    a, b = systems.lig50(seed=1), systems.lig50(seed=1)
    assert np.array_equal(a.pos, b.pos) and a.n_atoms == 50 and a.bond_idx.shape == (49, 2)
    assert not np.array_equal(a.pos, systems.lig50(seed=2).pos)
    assert abs(float(a.charge.sum())) < 1e-5
    d = np.linalg.norm(a.pos[:, None] - a.pos[None], axis=-1) + np.eye(50) * 9
    assert d.min() > 1.3


def test_water_box_counts_and_geometry():
    s = systems.water_box(4, seed=3, jitter=0.0)
    assert s.n_atoms == 192 and s.bond_idx.shape == (128, 2) and s.angle_idx.shape == (64, 3)
    oh = np.linalg.norm(s.pos[s.bond_idx[:, 0]] - s.pos[s.bond_idx[:, 1]], axis=1)
    assert np.allclose(oh, 0.9572, atol=1e-4)
    assert abs(float(s.charge.sum())) < 1e-3
    assert s.excl_offsets[-1] == 2 * 192 // 3 * 3 and s.periodic
    assert np.allclose(s.box_hi, 4 * 3.1034)
    # velocities: zero net momentum
    p = (s.vel * s.mass[:, None]).sum(0)
    assert np.abs(p).max() < 1e-2


def test_dhfr23k_shape():
    s = systems.dhfr23k()
    assert s.n_atoms == 23558 and s.periodic and np.allclose(s.box_hi, 62.23)
    assert s.bond_idx.max() < s.n_atoms and s.dihedral_idx.shape[0] > 2000
    n_chain = 2489
    t_per_atom = (np.sum((s.bond_idx < n_chain).all(1)) / n_chain, np.sum((s.angle_idx < n_chain).all(1)) / n_chain)
    assert 0.9 < t_per_atom[0] < 1.1 and 1.2 < t_per_atom[1] < 2.4
    # nothing overlapping: solute-water clearance
    from scipy.spatial import cKDTree
    d, _ = cKDTree(s.pos[:n_chain]).query(s.pos[n_chain::3])
    assert d.min() > 2.0


def test_to_c_aliases_arrays():
    s = systems.lig50()
    c = s.to_c()
    assert c.n_atoms == 50 and c.n_bonds == 49 and c.periodic == 0
    assert c.pos[3] == pytest.approx(float(s.pos[1, 0]))
    assert MdConfig().to_c().coulomb_k == pytest.approx(332.0637)
