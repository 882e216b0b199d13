"""Synthetic inputs: drug-like feature molecules and packed libraries.

Nothing here mirrors reference code; the reference ships no test data generator
and its example library is a missing blob. Two generators:

* `random_molecule` - a small molecular graph (rings, chains, charged and polar groups) with the
  pharmacophore features OpenBabel perception would report for it (`ligand_utils.py:25-88` decides
  them from atom predicates; here the predicates are drawn, not perceived). Output is a
  `LigandFeatures`, the input of the graph builder, so the same molecule can be fed to the
  reference's `LigandGraph` (golden fixtures) and to `library.cluster_ligand`.
* `synthetic_library` - a packed library of N such ligands for benchmarks, drawn per ligand from a
  counter-based stream (`seed`, ligand index) so any shard of it can be produced independently.
"""

from __future__ import annotations

import math

import numpy as np

from pharmaconet_amd.constants import TYPE_ID
from pharmaconet_amd.library import LigandFeatures, PackedLibrary, pack_ligand

__all__ = ["random_molecule", "synthetic_library", "ligand_rng"]

BASE_SEED = 20250523
C, N, O, F, P, S, CL = 6, 7, 8, 9, 15, 16, 17


class _Mol:
    def __init__(self):
        self.z: list[int] = []
        self.nbrs: list[list[int]] = []
        self.aromatic: list[bool] = []
        self.rings: list[tuple[int, ...]] = []
        self.hbd: set[int] = set()
        self.hba: set[int] = set()
        self.cation: list[tuple[int | tuple[int, ...], int | tuple[int, ...]]] = []
        self.anion: list[tuple[int | tuple[int, ...], int | tuple[int, ...]]] = []
        self.halogen: list[int] = []
        self.ring_of: dict[int, int] = {}

    def add_atom(self, z: int, aromatic: bool = False) -> int:
        self.z.append(z)
        self.nbrs.append([])
        self.aromatic.append(aromatic)
        return len(self.z) - 1

    def bond(self, a: int, b: int) -> None:
        self.nbrs[a].append(b)
        self.nbrs[b].append(a)

    def free_carbons(self) -> list[int]:
        out = []
        for i, z in enumerate(self.z):
            if z != C:
                continue
            cap = 3 if self.aromatic[i] else 4
            if len(self.nbrs[i]) < cap:
                out.append(i)
        return out


def _attach_point(mol: _Mol, rng: np.random.Generator) -> int | None:
    free = mol.free_carbons()
    if not free:
        return None
    return int(free[rng.integers(len(free))])


def _add_ring(mol: _Mol, rng, hetero: bool) -> int:
    size = 6 if rng.random() < 0.8 else 5
    atoms = [mol.add_atom(C, aromatic=True) for _ in range(size)]
    if hetero:
        k = int(rng.integers(size))
        mol.z[atoms[k]] = N
        mol.hba.add(atoms[k])
    for i in range(size):
        mol.bond(atoms[i], atoms[(i + 1) % size])
    rid = len(mol.rings)
    mol.rings.append(tuple(sorted(atoms)))
    for a in atoms:
        mol.ring_of[a] = rid
    carbons = [a for a in atoms if mol.z[a] == C]
    return int(carbons[rng.integers(len(carbons))])


def _add_fragment(mol: _Mol, rng, kind: str) -> int | None:
    """Adds a fragment, returns the atom through which it bonds to the rest (or None)."""
    if kind == "benzene":
        return _add_ring(mol, rng, hetero=False)
    if kind == "pyridine":
        return _add_ring(mol, rng, hetero=True)
    if kind == "chain":
        length = int(rng.integers(1, 5))
        prev = first = mol.add_atom(C)
        for _ in range(length - 1):
            nxt = mol.add_atom(C)
            mol.bond(prev, nxt)
            prev = nxt
        return first
    if kind == "tbutyl":
        center = mol.add_atom(C)
        for _ in range(3):
            mol.bond(center, mol.add_atom(C))
        return center
    if kind == "hydroxyl":
        o = mol.add_atom(O)
        mol.hbd.add(o)
        mol.hba.add(o)
        return o
    if kind == "ether":
        o = mol.add_atom(O)
        mol.hba.add(o)
        c = mol.add_atom(C)
        mol.bond(o, c)
        return o
    if kind == "carbonyl":
        c = mol.add_atom(C)
        o = mol.add_atom(O)
        mol.bond(c, o)
        mol.hba.add(o)
        return c
    if kind == "amine":
        n = mol.add_atom(N)
        mol.hbd.add(n)
        if rng.random() < 0.5:
            mol.hba.add(n)
        return n
    if kind == "amide":
        c = mol.add_atom(C)
        o = mol.add_atom(O)
        n = mol.add_atom(N)
        mol.bond(c, o)
        mol.bond(c, n)
        mol.hba.add(o)
        mol.hbd.add(n)
        return c
    if kind == "tert_amine":
        n = mol.add_atom(N)
        for _ in range(2):
            mol.bond(n, mol.add_atom(C))
        mol.cation.append((n, n))
        if rng.random() < 0.6:
            mol.hba.add(n)  # same atom key -> merged node types [Cation, HBond_acceptor]
        return n
    if kind == "guanidine":
        c = mol.add_atom(C)
        ns = [mol.add_atom(N) for _ in range(3)]
        for n in ns:
            mol.bond(c, n)
            mol.hbd.add(n)
        mol.cation.append(((c, *ns), c))
        return ns[0]
    if kind == "carboxylate":
        c = mol.add_atom(C)
        o1, o2 = mol.add_atom(O), mol.add_atom(O)
        mol.bond(c, o1)
        mol.bond(c, o2)
        mol.hba.add(o1)
        mol.hba.add(o2)
        mol.anion.append(((c, o1, o2), (o1, o2)))
        return c
    if kind == "phosphate":
        p = mol.add_atom(P)
        os_ = [mol.add_atom(O) for _ in range(4)]
        for o in os_:
            mol.bond(p, o)
        for o in os_[1:]:
            mol.hba.add(o)
        mol.anion.append(((p, *os_), p))
        return os_[0]
    if kind == "halogen":
        x = mol.add_atom(F if rng.random() < 0.5 else CL)
        mol.halogen.append(x)
        return x
    raise ValueError(kind)


_FRAGMENTS = (
    ("benzene", 0.16),
    ("pyridine", 0.06),
    ("chain", 0.22),
    ("tbutyl", 0.04),
    ("hydroxyl", 0.08),
    ("ether", 0.06),
    ("carbonyl", 0.07),
    ("amine", 0.07),
    ("amide", 0.07),
    ("tert_amine", 0.05),
    ("guanidine", 0.02),
    ("carboxylate", 0.04),
    ("phosphate", 0.01),
    ("halogen", 0.05),
)


def _topology(rng: np.random.Generator, n_fragments: int) -> _Mol:
    mol = _Mol()
    names = [f[0] for f in _FRAGMENTS]
    probs = np.array([f[1] for f in _FRAGMENTS])
    probs = probs / probs.sum()
    start = "benzene" if rng.random() < 0.6 else "chain"
    _add_fragment(mol, rng, start)
    for _ in range(n_fragments - 1):
        kind = names[int(rng.choice(len(names), p=probs))]
        anchor = _attach_point(mol, rng)
        if anchor is None:
            break
        port = _add_fragment(mol, rng, kind)
        if port is not None:
            mol.bond(anchor, port)
    return mol


def _features(mol: _Mol) -> list[tuple[str, int | tuple[int, ...], int | tuple[int, ...]]]:
    """The feature list in the reference's type order (`ligand_utils.py:80-88`)."""
    feats: list[tuple[str, int | tuple[int, ...], int | tuple[int, ...]]] = []
    for i, z in enumerate(mol.z):  # ligand_utils.py:36-40
        if z == C and all(mol.z[j] == C for j in mol.nbrs[i]):
            feats.append(("Hydrophobic", i, i))
    for ring in sorted(mol.rings):  # ligand_utils.py:47-52
        feats.append(("Aromatic", ring, ring))
    for atoms, center in mol.cation:
        feats.append(("Cation", atoms, center))
    for atoms, center in mol.anion:
        feats.append(("Anion", atoms, center))
    for i in sorted(mol.hbd):
        feats.append(("HBond_donor", i, i))
    for i in sorted(mol.hba):
        feats.append(("HBond_acceptor", i, i))
    for i in sorted(mol.halogen):
        if any(mol.z[j] == C for j in mol.nbrs[i]):
            feats.append(("Halogen", i, i))
    return feats


def _unit(rng) -> np.ndarray:
    v = rng.normal(size=3)
    return v / (np.linalg.norm(v) + 1e-12)


def _embed_random_walk(mol: _Mol, rng, origin: np.ndarray, step: float = 1.5) -> np.ndarray:
    n = len(mol.z)
    pos = np.zeros((n, 3))
    placed = np.zeros(n, dtype=bool)
    drift = _unit(rng)
    for root in range(n):
        if placed[root]:
            continue
        pos[root] = origin + rng.normal(scale=2.0, size=3)
        placed[root] = True
        queue = [root]
        while queue:
            a = queue.pop(0)
            for b in mol.nbrs[a]:
                if placed[b]:
                    continue
                direction = _unit(rng) + 0.8 * drift
                direction /= np.linalg.norm(direction) + 1e-12
                pos[b] = pos[a] + step * direction
                placed[b] = True
                queue.append(b)
    # make rings compact: pull ring atoms onto a circle around their centroid
    for ring in mol.rings:
        idx = list(ring)
        center = pos[idx].mean(axis=0)
        u = _unit(rng)
        w = np.cross(u, _unit(rng))
        w /= np.linalg.norm(w) + 1e-12
        for k, a in enumerate(idx):
            ang = 2 * math.pi * k / len(idx)
            pos[a] = center + 1.4 * (math.cos(ang) * u + math.sin(ang) * w)
    return pos


def _embed_on_model(mol: _Mol, feats, rng, model_nodes: tuple[np.ndarray, np.ndarray], spread: float) -> np.ndarray:
    """Place the features near type-compatible model nodes so that the ligand resembles an active."""
    centers, types = model_nodes
    pos = _embed_random_walk(mol, rng, centers.mean(axis=0))
    used: set[int] = set()
    order = list(range(len(feats)))
    rng.shuffle(order)
    fixed = np.zeros(len(mol.z), dtype=bool)
    for fi in order:
        ftype, atoms, center = feats[fi]
        compatible = [m for m in range(len(types)) if types[m] == TYPE_ID[ftype] and m not in used]
        if not compatible or rng.random() < 0.15:
            continue
        m = int(compatible[rng.integers(len(compatible))])
        used.add(m)
        target = centers[m] + rng.normal(scale=spread, size=3)
        idx = [atoms] if isinstance(atoms, int) else list(atoms)
        cidx = [center] if isinstance(center, int) else list(center)
        if any(fixed[a] for a in idx):
            continue
        shift = target - pos[cidx].mean(axis=0)
        for a in idx:
            pos[a] = pos[a] + shift
            fixed[a] = True
    return pos


def random_molecule(
    rng: np.random.Generator,
    num_conformers: int = 8,
    n_fragments: int | None = None,
    model_nodes: tuple[np.ndarray, np.ndarray] | None = None,
    active_like: bool = False,
    conformer_noise: float = 0.45,
    origin: np.ndarray | None = None,
) -> LigandFeatures:
    """Draw one feature molecule. With `active_like` and `model_nodes = (centers [Nm,3], type ids [Nm])`
    the features are laid over compatible model nodes; otherwise coordinates are a random walk."""
    if n_fragments is None:
        n_fragments = int(np.clip(round(rng.normal(6.0, 2.5)), 1, 14))
    mol = _topology(rng, n_fragments)
    feats = _features(mol)
    if origin is None:
        origin = np.zeros(3) if model_nodes is None else model_nodes[0].mean(axis=0)
    if active_like and model_nodes is not None:
        base = _embed_on_model(mol, feats, rng, model_nodes, spread=0.6)
    else:
        base = _embed_random_walk(mol, rng, origin)
    n = len(mol.z)
    noise = rng.normal(scale=conformer_noise, size=(n, num_conformers, 3))
    atom_positions = (base[:, None, :] + noise).astype(np.float32)
    return LigandFeatures(
        atomic_nums=list(mol.z),
        heavy_neighbors=[list(x) for x in mol.nbrs],
        features=feats,
        atom_positions=atom_positions,
    )


def ligand_rng(seed: int, index: int) -> np.random.Generator:
    """Counter-based stream: ligand `index` of library `seed` draws from its own Philox key."""
    return np.random.Generator(np.random.Philox(key=[int(seed) & (2**64 - 1), int(index)]))


def synthetic_library(
    count: int,
    first: int = 0,
    num_conformers: int = 8,
    model_nodes: tuple[np.ndarray, np.ndarray] | None = None,
    active_fraction: float = 0.1,
    seed: int = BASE_SEED,
    max_nodes: int = 32,
    conformer_noise: float = 0.45,
    molecules_out: list | None = None,
) -> PackedLibrary:
    """Ligands `first .. first + count` of the synthetic library `seed` (molecule-level generator).

    Ligands whose graph exceeds `max_nodes` pharmacophore nodes are redrawn with fewer fragments so
    that the library respects the `<= 32 pharmacophore points` shape of BASELINE.json's configs."""
    records: list[bytes] = []
    for index in range(first, first + count):
        rng = ligand_rng(seed, index)
        active = model_nodes is not None and rng.random() < active_fraction
        n_fragments = None
        while True:
            lig = random_molecule(
                rng, num_conformers, n_fragments=n_fragments, model_nodes=model_nodes, active_like=active,
                conformer_noise=conformer_noise,
            )
            rec = pack_ligand(lig)
            n_nodes = int.from_bytes(rec[0:2], "little")
            if n_nodes <= max_nodes:
                break
            n_fragments = max(1, (n_fragments or 8) - 2)
        records.append(rec)
        if molecules_out is not None:
            molecules_out.append(lig)
    return PackedLibrary.from_records(records)


# ------------------------------------------------------------------ device-side library expansion
def coordinate_layout(lib: PackedLibrary) -> tuple[np.ndarray, np.ndarray, int]:
    """Per 4-byte word of `lib.data`: is it a coordinate, and which (ligand node, axis) it belongs to.

    Returns `(is_coord bool [W], group int32 [W], n_groups)`; words of one node's x (or y, z) over all
    conformers share a group id, so a per-group offset moves that node rigidly in every conformer."""
    n_words = int(lib.data.shape[0]) // 4
    is_coord = np.zeros(n_words, dtype=bool)
    group = np.zeros(n_words, dtype=np.int32)
    next_group = 0
    hdr = lib.headers()
    for i in range(len(lib)):
        n, c = int(hdr[i, 0]), int(hdr[i, 1])
        k = int(hdr[i, 2])
        start = int(lib.offsets[i]) + ((8 + n + k + 3) & ~3)
        w0 = start // 4
        count = 3 * n * c
        is_coord[w0 : w0 + count] = True
        group[w0 : w0 + count] = next_group + np.repeat(np.arange(3 * n, dtype=np.int32), c)
        next_group += 3 * n
    return is_coord, group, next_group


def expand_library_on_device(
    base: PackedLibrary,
    replicas: int,
    device,
    seed: int = BASE_SEED,
    ligand_sigma: float = 0.35,
    conformer_sigma: float = 0.30,
):
    """`replicas` perturbed copies of every ligand of `base`, built in HBM with torch.

    Copy r of base ligand b keeps b's topology (types, clusters) and gets its own geometry: every node is
    displaced by N(0, ligand_sigma) (the same in all conformers) and every conformer coordinate by a
    further N(0, conformer_sigma). Ligand index = r * len(base) + b. Returns `(offsets int64 [N + 1],
    data uint8)` device tensors in the packed library format."""
    import torch

    dev = torch.device(device)
    is_coord, group, n_groups = coordinate_layout(base)
    base_words = torch.from_numpy(base.data.view(np.float32).copy()).to(dev)
    coord = torch.from_numpy(is_coord).to(dev)
    gid = torch.from_numpy(group.astype(np.int64)).to(dev)
    n_words = base_words.numel()
    nbytes = n_words * 4
    out = torch.empty((replicas, n_words), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    for r in range(replicas):
        node_shift = torch.randn(n_groups, generator=gen, device=dev, dtype=torch.float32) * ligand_sigma
        noise = torch.randn(n_words, generator=gen, device=dev, dtype=torch.float32) * conformer_sigma
        out[r] = torch.where(coord, base_words + node_shift[gid] + noise, base_words)
    base_off = torch.from_numpy(base.offsets[:-1].astype(np.int64)).to(dev)
    offsets = (base_off[None, :] + (torch.arange(replicas, device=dev, dtype=torch.int64) * nbytes)[:, None]).reshape(-1)
    offsets = torch.cat([offsets, torch.tensor([replicas * nbytes], device=dev, dtype=torch.int64)])
    return offsets, out.view(torch.uint8).reshape(-1)
