"""`energy_potential_between_mols` of the CPU oracle (oracle/mdx_oracle.c: orc_between_mols) - the checker of
tests/test_gpu_between_mols.py - against closed forms and its own total energies.  The reference consumes the matrix at
/root/reference src/properties/crystal.rs:347-370, 533; its arithmetic lives in the absent `dynamics` crate (parity unpinned)."""
import numpy as np

from molchanica_amd import MdConfig, MdSystem, systems
from oracle import oracle


def groups_by_molecule(s):
    ms = np.asarray(s.mol_start, dtype=np.int64)
    g = np.zeros(s.n_atoms, np.uint8)
    for m in range(len(ms)):
        g[ms[m]:(ms[m + 1] if m + 1 < len(ms) else s.n_atoms)] = m
    return g, len(ms)


def test_two_diatomics_closed_form():
    # molecules A = (0, 1), B = (2, 3) on a line in vacuum; 1-2 pairs excluded; the matrix element is the four cross pairs
    x = np.array([[0, 0, 0], [1, 0, 0], [4, 0, 0], [5, 0, 0]], dtype=np.float32)
    q = np.array([0.3, -0.3, 0.5, -0.5], dtype=np.float32)
    s = MdSystem(pos=x, mass=np.full(4, 12.0), charge=q, lj_type=np.zeros(4, np.uint32), lj_sigma=[3.0], lj_eps=[0.1],
                 bond_idx=[[0, 1], [2, 3]], bond_k=[300.0, 300.0], bond_r0=[1.0, 1.0],
                 excl_offsets=[0, 1, 2, 3, 4], excl_idx=[1, 0, 3, 2], mol_start=[0, 2]).normalise()
    cfg = MdConfig(lj_cutoff=0.0, coulomb_cutoff=0.0)
    g, n = groups_by_molecule(s)
    m, gross = oracle.between_mols(s, cfg, g, n)
    want = 0.0
    eps, ke = float(np.float32(0.1)), float(np.float32(332.0637))      # the f32 values the ABI carries
    for i in (0, 1):
        for j in (2, 3):
            r = abs(float(x[j, 0] - x[i, 0]))
            want += 4 * eps * ((3.0 / r) ** 12 - (3.0 / r) ** 6) + ke * float(q[i]) * float(q[j]) / r
    assert abs(m[0, 1] - want) < 1e-9 * max(1.0, abs(want))
    assert m[0, 0] == 0.0 and m[1, 1] == 0.0 and m[1, 0] == m[0, 1]
    assert gross[0, 1] >= abs(m[0, 1])


def test_matrix_adds_up_to_the_nonbonded_energy_and_cells_agree():
    s = systems.molecular_crystal()
    cfg = MdConfig(lj_cutoff=8.0, coulomb_cutoff=8.0, skin=1.0)
    g, n = groups_by_molecule(s)
    m, _ = oracle.between_mols(s, cfg, g, n)
    m2, _ = oracle.between_mols(s, cfg, g, n, use_cells=True)
    _, e = oracle.forces(s, cfg)
    iu = np.triu_indices(n)
    assert abs(m[iu].sum() - (e["lj"] + e["coulomb"] + e["lj14"] + e["coulomb14"])) < 1e-9 * abs(e["potential_nonbonded"])
    assert np.abs(m - m2).max() < 1e-10 and np.array_equal(m, m.T)
    # the cohesive energy the reference derives from it (crystal.rs:347-370): sum of the strict upper triangle / n_mol
    off = m[np.triu_indices(n, 1)].sum()
    assert abs(off - (m[iu].sum() - np.trace(m))) < 1e-9


def test_group_map_coarsens_the_molecule_matrix():
    s = systems.small_complex()
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1)
    ms = np.asarray(s.mol_start)
    g3 = np.full(s.n_atoms, 2, np.uint8); g3[:ms[1]] = 0; g3[ms[1]:ms[2]] = 1      # receptor / ligand / solvent
    m3, _ = oracle.between_mols(s, cfg, g3, 3, use_cells=True)
    _, e = oracle.forces(s, cfg, use_cells=True)
    assert abs(m3[np.triu_indices(3)].sum() - e["potential_nonbonded"]) < 1e-8 * abs(e["potential_nonbonded"])
    assert m3[0, 1] != 0.0 and m3[1, 2] != 0.0
