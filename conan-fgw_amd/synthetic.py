"""Synthetic conformer batches shaped like the reference's MoleculeNet inputs.

The reference ships no datasets (``/data`` is git-ignored, README.md:90-92), so every test and the
benchmark run on random-coordinate conformers.  The layout reproduces what the reference's
``collate_fn`` hands to the model (conan_fgw/src/data/datasets.py:170-199): one flat batch of
``G = B*K`` conformer graphs, molecule-major, the K conformers of a molecule sharing ``n`` and ``z``
and differing only in ``pos``; ``batch`` is the conformer-graph id of each atom (sorted).

Pure host-side numpy; no GPU, no oracle.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

# element distribution of SURVEY.md section 8(d): explicit hydrogens, never z == 0
_Z_VALUES = np.array([1, 6, 7, 8, 9, 16, 17], dtype=np.int64)
_Z_PROBS = np.array([0.50, 0.30, 0.06, 0.09, 0.01, 0.02, 0.02])

# (mean, std, lo, hi) of atoms per conformer; SURVEY.md section 8(d) / BASELINE.md section 3
SHAPES = {
    "esol": (20.0, 6.0, 6, 33),
    "lipo": (48.0, 12.0, 15, 110),
    "bace": (65.0, 12.0, 30, 120),
    "freesolv": (18.0, 6.0, 4, 33),
}

# BASELINE.json configs -> (shape, B, K)
CONFIGS = {
    "cfg1": ("esol", 32, 5),
    "cfg2": ("esol", 256, 5),
    "cfg3": ("lipo", 1024, 5),   # sharded over 8 GPUs: 128 molecules per rank
    "cfg4": ("bace", 64, 5),
    "cfg5": ("freesolv", 64, 20),
}


@dataclasses.dataclass
class ConformerBatch:
    """Flat molecule-major batch of B*K conformer graphs."""

    z: np.ndarray          # [sumN] int64 atomic numbers
    pos: np.ndarray        # [sumN, 3] float32
    batch: np.ndarray      # [sumN] int64 conformer-graph id (sorted, 0..G-1)
    y: np.ndarray          # [B] float32 regression target
    num_molecules: int
    num_conformers: int
    atoms_per_molecule: np.ndarray   # [B] int64

    @property
    def num_graphs(self) -> int:
        return self.num_molecules * self.num_conformers

    @property
    def max_nodes(self) -> int:
        return int(self.atoms_per_molecule.max())

    @property
    def graph_ptr(self) -> np.ndarray:
        n = np.repeat(self.atoms_per_molecule, self.num_conformers)
        return np.concatenate([[0], np.cumsum(n)]).astype(np.int64)


def _sample_conformer(rng: np.random.RandomState, n: int, box: float, min_dist: float) -> np.ndarray:
    pos = rng.uniform(0.0, box, size=(n, 3))
    for _ in range(200):
        diff = pos[:, None, :] - pos[None, :, :]
        d2 = (diff * diff).sum(-1) + np.eye(n) * 1e9
        bad = np.unique(np.nonzero(d2 < min_dist * min_dist)[0])
        if bad.size == 0:
            break
        pos[bad] = rng.uniform(0.0, box, size=(bad.size, 3))
    return pos.astype(np.float32)


def make_batch(
    shape: str = "esol",
    num_molecules: int = 32,
    num_conformers: int = 5,
    seed: int = 1234,
    box: Optional[float] = None,
    density: float = 0.10,
    min_dist: float = 0.7,
    fixed_atoms: Optional[int] = None,
) -> ConformerBatch:
    """Draw a batch.  ``box=None`` uses L=(n/density)^(1/3) Angstrom per molecule (all pairs well inside
    10 A for ESOL-sized molecules); ``box=16.0`` is the "stretched" variant used by the neighbour-exactness
    tests so that the radius cutoff is actually exercised.  ``fixed_atoms`` pins every molecule to n atoms."""
    mean, std, lo, hi = SHAPES[shape]
    rng = np.random.RandomState(seed)
    zs, poss, batches, ns = [], [], [], []
    g = 0
    for _ in range(num_molecules):
        n = int(fixed_atoms) if fixed_atoms is not None else int(np.clip(np.rint(rng.normal(mean, std)), lo, hi))
        z = rng.choice(_Z_VALUES, size=n, p=_Z_PROBS)
        L = float(box) if box is not None else (n / density) ** (1.0 / 3.0)
        for _k in range(num_conformers):
            zs.append(z)
            poss.append(_sample_conformer(rng, n, L, min_dist))
            batches.append(np.full(n, g, dtype=np.int64))
            g += 1
        ns.append(n)
    y = rng.normal(size=num_molecules).astype(np.float32)
    return ConformerBatch(
        z=np.concatenate(zs).astype(np.int64),
        pos=np.concatenate(poss).astype(np.float32),
        batch=np.concatenate(batches),
        y=y,
        num_molecules=num_molecules,
        num_conformers=num_conformers,
        atoms_per_molecule=np.asarray(ns, dtype=np.int64),
    )


def make_config(name: str, num_molecules: Optional[int] = None, seed_offset: int = 0) -> ConformerBatch:
    """Batch for one of BASELINE.json's configs (seed 1234 + config index)."""
    shape, B, K = CONFIGS[name]
    idx = int(name[3:])
    return make_batch(shape, num_molecules or B, K, seed=1234 + idx + 1000 * seed_offset)


@dataclasses.dataclass
class BondBatch:
    """2-D (covalent) graph of the same batch: what the reference's featurisation puts into `batch.x`, `batch.edge_index`,
    `batch.edge_attr` (datasets.py: 9 integer atom features, 3 integer bond features, both directions of every bond)."""
    x: np.ndarray           # [sumN, 9] float32
    edge_index: np.ndarray  # [2, E] int64 (row 0 = source, row 1 = target), global atom ids
    edge_attr: np.ndarray   # [E, 3] float32


def make_bond_graph(b: ConformerBatch, seed: int = 0) -> BondBatch:
    """Random molecular-looking bond graphs: per molecule a random tree (every atom bonded to an earlier one) plus a few
    ring-closing bonds; the K conformers of a molecule share the 2-D graph and its features.  Edges are emitted in a shuffled
    order (the kernels must not rely on any ordering of `edge_index`)."""
    rng = np.random.RandomState(seed)
    K = b.num_conformers
    xs, srcs, dsts, attrs = [], [], [], []
    base = 0
    for n in b.atoms_per_molecule:
        n = int(n)
        feat = rng.randint(0, 6, size=(n, 9)).astype(np.float32)
        bonds = [(i, int(rng.randint(0, i))) for i in range(1, n)]
        for _ in range(max(0, n // 8)):
            i, j = int(rng.randint(0, n)), int(rng.randint(0, n))
            if i != j and (i, j) not in bonds and (j, i) not in bonds:
                bonds.append((i, j))
        battr = rng.randint(0, 4, size=(len(bonds), 3)).astype(np.float32)
        for _k in range(K):
            xs.append(feat)
            for (i, j), a in zip(bonds, battr):
                srcs += [base + i, base + j]; dsts += [base + j, base + i]; attrs += [a, a]
            base += n
    perm = rng.permutation(len(srcs))
    ei = np.stack([np.asarray(srcs, dtype=np.int64)[perm], np.asarray(dsts, dtype=np.int64)[perm]])
    ea = np.asarray(attrs, dtype=np.float32).reshape(-1, 3)[perm]
    return BondBatch(np.concatenate(xs) if xs else np.zeros((0, 9), np.float32), ei, ea)
