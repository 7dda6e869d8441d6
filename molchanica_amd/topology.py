"""Bonded-term and exclusion lists from a bond graph.

Host-side system build (the part of `MdState::new` that "build[s] bonded term lists from
bonds/adjacency"; input shape at /root/reference src/md/mod.rs:1110-1120
`bonds`, `adjacency_list`).  Setup-time only; not on the accelerated path.
"""
from __future__ import annotations

import numpy as np


def adjacency(n_atoms: int, bonds: np.ndarray) -> list[list[int]]:
    adj: list[list[int]] = [[] for _ in range(n_atoms)]
    for a, b in np.asarray(bonds, dtype=np.int64).reshape(-1, 2):
        adj[a].append(int(b))
        adj[b].append(int(a))
    return [sorted(set(x)) for x in adj]


def angles_from_bonds(adj: list[list[int]]) -> np.ndarray:
    out = []
    for j, nb in enumerate(adj):
        for x in range(len(nb)):
            for y in range(x + 1, len(nb)):
                out.append((nb[x], j, nb[y]))
    return np.asarray(out, dtype=np.uint32).reshape(-1, 3)


def dihedrals_from_bonds(adj: list[list[int]]) -> np.ndarray:
    out = []
    for j, nbj in enumerate(adj):
        for k in nbj:
            if k <= j:
                continue
            for i in nbj:
                if i == k:
                    continue
                for l in adj[k]:
                    if l == j or l == i:
                        continue
                    out.append((i, j, k, l))
    return np.asarray(out, dtype=np.uint32).reshape(-1, 4)


def exclusions_and_pairs14(n_atoms: int, adj: list[list[int]]):
    """1-2 and 1-3 partners -> exclusion CSR (symmetric); 1-4 partners that are not also
    1-2/1-3 -> unique (i<j) pair list."""
    excl: list[set[int]] = [set() for _ in range(n_atoms)]
    for i, nb in enumerate(adj):
        for j in nb:
            excl[i].add(j)
            for k in adj[j]:
                if k != i:
                    excl[i].add(k)
    p14 = set()
    for i, nb in enumerate(adj):
        for j in nb:
            for k in adj[j]:
                if k == i:
                    continue
                for l in adj[k]:
                    if l == j or l == i or l in excl[i]:
                        continue
                    p14.add((min(i, l), max(i, l)))
    offsets = np.zeros(n_atoms + 1, dtype=np.uint32)
    for i in range(n_atoms):
        offsets[i + 1] = offsets[i] + len(excl[i])
    idx = np.zeros(int(offsets[-1]), dtype=np.uint32)
    for i in range(n_atoms):
        idx[offsets[i]:offsets[i + 1]] = sorted(excl[i])
    pairs = np.asarray(sorted(p14), dtype=np.uint32).reshape(-1, 2)
    return offsets, idx, pairs


def csr_from_pairs(n_atoms: int, pairs: np.ndarray):
    """Symmetric exclusion CSR from an (i,j) pair array — vectorised, for large systems."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    both = np.concatenate([pairs, pairs[:, ::-1]], axis=0)
    order = np.lexsort((both[:, 1], both[:, 0]))
    both = both[order]
    counts = np.bincount(both[:, 0], minlength=n_atoms)
    offsets = np.zeros(n_atoms + 1, dtype=np.uint32)
    offsets[1:] = np.cumsum(counts)
    return offsets, both[:, 1].astype(np.uint32)
