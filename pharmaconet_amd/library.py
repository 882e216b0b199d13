"""Packed ligand library: the byte format the HIP scoring kernel streams from HBM.

One *record* per ligand holds exactly the numeric data `GraphMatcher.run()` reads from a
`LigandGraph` (reference `src/pmnet/scoring/ligand.py:110-259`, `graph_match.py:43-172`):

* the pharmacophore nodes - a 7-bit type mask each (`LigandNode.types`, `ligand.py:137-140,280`)
  and a float32 position per conformer (`LigandNode.set_positions`, `ligand.py:293-301`);
* the node clusters, already in `priority_fn` order (`graph_match.py:43-60,87`), each a
  contiguous range of the node list in cluster iteration order (`LigandNodeCluster.__iter__`,
  `ligand.py:387-395`: the high-priority node first). The sort key does not depend on the
  pharmacophore model, so sorting happens once, here; the model-dependent steps (dropping
  clusters without a candidate, the depth cap of 20 - `graph_match.py:124-137,88`) run on the GPU.

Record layout (little endian, every record starts on a 16-byte boundary):

    u16 n_nodes | u16 n_conf | u16 n_clusters | u16 0
    u8  typemask[n_nodes]
    u8  cluster_end[n_clusters]        exclusive end of cluster i in the node list
    pad to 4 bytes
    f32 xyz[n_nodes][3][n_conf]        conformer index fastest
    pad to 16 bytes

A library is `offsets: u64[N + 1]` (byte offsets into `data`) plus `data: u8[...]`.
"""

from __future__ import annotations

import struct
from collections.abc import Iterable, Sequence
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .constants import (
    CLUSTER_PRIORITY,
    MAX_CONFORMERS,
    MAX_LIGAND_CLUSTERS,
    MAX_LIGAND_NODES,
    TYPE_ID,
)

__all__ = [
    "LigandFeatures",
    "ClusteredLigand",
    "PackedLibrary",
    "cluster_ligand",
    "pack_clustered_ligand",
    "pack_ligand",
    "pack_ligand_or_marker",
    "pack_features_native",
    "flatten_features",
    "UNSUPPORTED_RECORD",
    "as_packed_library",
]

_HEADER = struct.Struct("<HHHH")
RECORD_ALIGN = 16
_MAGIC = b"PMXLIB01"


# --------------------------------------------------------------------------- inputs
@dataclass
class LigandFeatures:
    """What ligand perception hands to the graph builder (`Ligand.__init__`, `ligand.py:16-61`).

    `features` is the reference's `pharmacophore_list`: `(type, atom_indices, center_indices)` in
    the type order of `ligand_utils.py:80-88`; `atom_indices` / `center_indices` are an int or a
    tuple of ints exactly as `PharmacophoreNode` holds them (an int and a 1-tuple are different
    keys, `ligand.py:137`). `heavy_neighbors[i]` lists the heavy-atom neighbours of atom i and
    `atomic_nums[i]` its element (what `ob.OBAtomAtomIter` / `GetAtomicNum` provide to
    `ligand.py:169-171,202-204`). `atom_positions` is float32 `[N_atoms, N_conformers, 3]`.
    """

    atomic_nums: Sequence[int]
    heavy_neighbors: Sequence[Sequence[int]]
    features: Sequence[tuple[str, int | tuple[int, ...], int | tuple[int, ...]]]
    atom_positions: np.ndarray

    @property
    def num_atoms(self) -> int:
        return len(self.atomic_nums)

    @property
    def num_conformers(self) -> int:
        return int(self.atom_positions.shape[1])


@dataclass
class ClusteredLigand:
    """Numeric content of a `LigandGraph` (before the priority sort)."""

    typemask: np.ndarray  # u8 [n]
    positions: np.ndarray  # f32 [n, C, 3]
    clusters: list[list[int]]  # node indices in cluster iteration order
    cluster_types: list[str]  # "Aromatic" | "Cation" | "Anion" | "HBond" | "Halogen" | "Hydrophobic"
    cluster_key_atom: list[int]  # min(cluster.nodes[0].atom_indices)  (graph_match.py:46)


# ------------------------------------------------------------- LigandGraph restated
@dataclass
class _Node:
    index: int
    types: list[str]
    atom_indices: frozenset[int]
    center_indices: int | tuple[int, ...]
    group: dict[int, None] = field(default_factory=dict)  # ordered set of node indices (may hold itself)
    dependence: set[int] = field(default_factory=set)


def _has_type(types: Iterable[str], *prefixes: str) -> bool:
    return any(t.startswith(prefixes) for t in types)


def cluster_ligand(lig: LigandFeatures) -> ClusteredLigand:
    """`LigandGraph.__init__` restated on plain data (`ligand.py:110-259`).

    Follows the reference step by step, quirks included: types are merged by the *key*
    `atom_indices` (`:137-140`); dependence rules are evaluated when the newer node is created,
    against the older node's types at that moment (`:151-152,311-328`); the hydrophobic flood pops
    start nodes last-in-first-out and may link a node to itself (`:194-213`); a low-priority node
    joins the cluster founded by a member of its group, else founds one (`:230-255`).
    """
    nodes: list[_Node] = []
    node_dict: dict[str, list[_Node]] = {}
    by_key: dict[int | tuple[int, ...], _Node] = {}

    # __add_nodes (ligand.py:134-156)
    for ftype, atom_indices, center_indices in lig.features:
        key = atom_indices if isinstance(atom_indices, int) else tuple(atom_indices)
        node = by_key.get(key)
        if node is not None:
            node.types.append(ftype)
            node_dict.setdefault(ftype, []).append(node)
            continue
        atoms = frozenset({key}) if isinstance(key, int) else frozenset(key)
        center = center_indices if isinstance(center_indices, int) else tuple(center_indices)
        new = _Node(len(nodes), [ftype], atoms, center)
        nodes.append(new)
        node_dict.setdefault(ftype, []).append(new)
        for old in nodes[:-1]:  # old.add_neighbors(new)  (ligand.py:303-329)
            if _has_type(old.types, "Hydrophobic") and _has_type(new.types, "Aromatic"):
                if old.atom_indices <= new.atom_indices:
                    old.dependence.add(new.index)
            elif _has_type(old.types, "Aromatic") and _has_type(new.types, "Hydrophobic"):
                if new.atom_indices <= old.atom_indices:
                    new.dependence.add(old.index)
            elif _has_type(old.types, "HBond") and _has_type(new.types, "Cation", "Anion"):
                if old.atom_indices <= new.atom_indices:
                    old.dependence.add(new.index)
            elif _has_type(old.types, "Cation", "Anion") and _has_type(new.types, "HBond"):
                if new.atom_indices <= old.atom_indices:
                    new.dependence.add(old.index)
        by_key[key] = new

    # __group_nodes, functional groups (ligand.py:158-192)
    hbond_groups: dict[int, list[_Node]] = {}
    hydrop_groups: dict[int, list[_Node]] = {}
    for node in nodes:
        if "HBond_acceptor" in node.types or "HBond_donor" in node.types:
            groups = hbond_groups
        elif "Hydrophobic" in node.types:
            groups = hydrop_groups
        else:
            continue
        assert len(node.atom_indices) == 1
        atom_index = next(iter(node.atom_indices))
        neighbors = [j for j in lig.heavy_neighbors[atom_index] if lig.atomic_nums[j] != 1]
        if len(neighbors) == 1:
            members = groups.setdefault(neighbors[0], [])
            for other in members:
                node.group[other.index] = None
                other.group[node.index] = None
            members.append(node)

    # __group_nodes, hydrophobic flood over carbon-carbon bonds (ligand.py:194-213)
    index_to_node = {next(iter(node.atom_indices)): node for node in node_dict.get("Hydrophobic", [])}
    while index_to_node:
        _, start = index_to_node.popitem()
        members = [start] + [nodes[i] for i in start.group]
        group_index = [next(iter(n.atom_indices)) for n in members]
        for atom_index in group_index:  # grows while iterating
            for nbr in lig.heavy_neighbors[atom_index]:
                if lig.atomic_nums[nbr] != 6:
                    continue
                reached = index_to_node.pop(nbr, None)
                if reached is None:
                    continue
                group_index.append(nbr)
                for n in members:
                    n.group[reached.index] = None
                    reached.group[n.index] = None
                members.append(reached)

    # __setup_cluster (ligand.py:215-259)
    in_cluster: set[int] = set()
    founder_cluster: dict[int, int] = {}  # node index -> cluster id (node_cluster_dict keys)
    clusters: list[list[int]] = []
    cluster_types: list[str] = []
    has_high: list[bool] = []

    def new_cluster(ctype: str) -> int:
        clusters.append([])
        cluster_types.append(ctype)
        has_high.append(False)
        return len(clusters) - 1

    for ftype in ("Aromatic", "Cation", "Anion", "Halogen"):
        for node in node_dict.get(ftype, []):
            if node.index in in_cluster:
                continue
            in_cluster.add(node.index)
            cid = new_cluster(ftype)
            clusters[cid].insert(0, node.index)  # high-priority node iterates first (ligand.py:387-395)
            has_high[cid] = True
            founder_cluster[node.index] = cid
    for ftype in ("Hydrophobic", "HBond_donor", "HBond_acceptor"):
        for node in node_dict.get(ftype, []):
            if node.index in in_cluster:
                continue
            in_cluster.add(node.index)
            add_new = True
            if node.dependence:
                clusters[founder_cluster[min(node.dependence)]].append(node.index)
                add_new = False
            elif node.group:
                for g in node.group:
                    if g in founder_cluster:
                        clusters[founder_cluster[g]].append(node.index)
                        add_new = False
                        break
            if add_new:
                cid = new_cluster("HBond" if ftype.startswith("HBond") else "Hydrophobic")
                clusters[cid].append(node.index)
                founder_cluster[node.index] = cid

    n = len(nodes)
    num_conf = lig.num_conformers
    typemask = np.zeros((n,), dtype=np.uint8)
    positions = np.zeros((n, num_conf, 3), dtype=np.float32)
    atom_positions = np.asarray(lig.atom_positions, dtype=np.float32)
    for node in nodes:
        mask = 0
        for t in node.types:
            mask |= 1 << TYPE_ID[t]
        typemask[node.index] = mask
        if isinstance(node.center_indices, int):  # LigandNode.set_positions (ligand.py:293-301)
            positions[node.index] = atom_positions[node.center_indices]
        else:
            positions[node.index] = np.mean(atom_positions[list(node.center_indices), :], axis=0, dtype=np.float32)
    key_atom = [min(nodes[c[0]].atom_indices) for c in clusters]
    return ClusteredLigand(typemask, positions, clusters, cluster_types, key_atom)


# ----------------------------------------------------------------------- packing
class LigandTooLarge(ValueError):
    """The ligand exceeds a structural limit of the GPU engine (include/pmx.h)."""


def pack_clustered_ligand(cl: ClusteredLigand) -> bytes:
    """Sort clusters by `priority_fn` (`graph_match.py:43-60`, stable like `sorted`, `:87`),
    renumber nodes cluster by cluster and emit one record."""
    order = sorted(
        range(len(cl.clusters)),
        key=lambda i: (
            CLUSTER_PRIORITY[cl.cluster_types[i]][0],
            -len(cl.clusters[i]),
            CLUSTER_PRIORITY[cl.cluster_types[i]][1],
            cl.cluster_key_atom[i],
        ),
    )
    node_order: list[int] = []
    cluster_end: list[int] = []
    for i in order:
        node_order.extend(cl.clusters[i])
        cluster_end.append(len(node_order))
    if len(set(node_order)) != len(node_order):
        raise ValueError("a ligand node may belong to one cluster only")
    n = len(node_order)
    positions = np.asarray(cl.positions, dtype=np.float32)
    num_conf = int(positions.shape[1]) if positions.ndim == 3 else 0
    if n > MAX_LIGAND_NODES:
        raise LigandTooLarge(f"ligand has {n} pharmacophore nodes; at most {MAX_LIGAND_NODES} are supported")
    if len(order) > MAX_LIGAND_CLUSTERS:
        raise LigandTooLarge(f"ligand has {len(order)} clusters; at most {MAX_LIGAND_CLUSTERS} are supported")
    if not 1 <= num_conf <= MAX_CONFORMERS:
        raise LigandTooLarge(f"ligand has {num_conf} conformers; between 1 and {MAX_CONFORMERS} are supported")
    head = _HEADER.pack(n, num_conf, len(order), 0)
    typemask = np.asarray(cl.typemask, dtype=np.uint8)[node_order].tobytes()
    ends = bytes(cluster_end)
    pad4 = (-(len(head) + len(typemask) + len(ends))) % 4
    xyz = np.ascontiguousarray(np.transpose(positions[node_order], (0, 2, 1))).tobytes()  # [n][3][C]
    body = head + typemask + ends + b"\0" * pad4 + xyz
    return body + b"\0" * ((-len(body)) % RECORD_ALIGN)


def pack_ligand(lig: LigandFeatures) -> bytes:
    return pack_clustered_ligand(cluster_ligand(lig))


def flatten_features(mols: Sequence[LigandFeatures]) -> dict[str, np.ndarray]:
    """A batch of `LigandFeatures` as the flat arrays of `pmx_feature_batch` (include/pmx.h)."""
    atom_off, feat_off, pos_off = [0], [0], [0]
    z, nbr_off, nbr = [], [0], []
    ftype, fflags, fa_off, fa, fc_off, fc = [], [], [0], [], [0], []
    n_conf, pos = [], []
    for m in mols:
        na = m.num_atoms
        z.extend(int(x) for x in m.atomic_nums)
        for row in m.heavy_neighbors:
            nbr.extend(int(j) for j in row)
            nbr_off.append(len(nbr))
        atom_off.append(atom_off[-1] + na)
        for t, atoms, centers in m.features:
            ftype.append(TYPE_ID[t])
            a_tuple, c_tuple = not isinstance(atoms, int), not isinstance(centers, int)
            fflags.append((1 if a_tuple else 0) | (2 if c_tuple else 0))
            fa.extend([int(atoms)] if not a_tuple else [int(x) for x in atoms])
            fa_off.append(len(fa))
            fc.extend([int(centers)] if not c_tuple else [int(x) for x in centers])
            fc_off.append(len(fc))
        feat_off.append(len(ftype))
        p = np.ascontiguousarray(m.atom_positions, dtype=np.float32)
        n_conf.append(int(p.shape[1]) if p.ndim == 3 else 0)
        pos.append(p.reshape(-1))
        pos_off.append(pos_off[-1] + p.size)
    return dict(
        atom_off=np.array(atom_off, np.uint64), atomic_num=np.array(z, np.uint8), nbr_off=np.array(nbr_off, np.uint64),
        nbr=np.array(nbr, np.int32), feat_off=np.array(feat_off, np.uint64), feat_type=np.array(ftype, np.uint8),
        feat_flags=np.array(fflags, np.uint8), feat_atom_off=np.array(fa_off, np.uint64), feat_atoms=np.array(fa, np.int32),
        feat_center_off=np.array(fc_off, np.uint64), feat_centers=np.array(fc, np.int32), n_conf=np.array(n_conf, np.int32),
        pos_off=np.array(pos_off, np.uint64), positions=np.concatenate(pos) if pos else np.zeros(0, np.float32),
    )


def pack_features_native(mols: Sequence[LigandFeatures] | dict[str, np.ndarray], threads: int = 1) -> tuple["PackedLibrary", np.ndarray]:
    """The packer in native code (`pmx_pack_features`, csrc/pmx_pack.cpp): byte-identical to `pack_ligand` per molecule,
    multi-threaded over molecules. Returns the library and a per-molecule status (1: outside the structural limits,
    packed as `UNSUPPORTED_RECORD`)."""
    import ctypes

    from . import _ffi

    flat = mols if isinstance(mols, dict) else flatten_features(mols)
    lib = _ffi.load_packer()  # host-only library: packing workers load no GPU runtime
    n = int(flat["atom_off"].shape[0]) - 1
    batch = _ffi.FeatureBatch(n, *(flat[k].ctypes.data for k in (
        "atom_off", "atomic_num", "nbr_off", "nbr", "feat_off", "feat_type", "feat_flags", "feat_atom_off", "feat_atoms",
        "feat_center_off", "feat_centers", "n_conf", "pos_off", "positions")))
    offsets = np.zeros(n + 1, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    nbytes = ctypes.c_uint64(0)
    def check(rc):
        if rc != 0:
            raise _ffi.PmxError(f"libpmx_pack error {rc}: {lib.pmx_last_error().decode()}")

    check(lib.pmx_pack_features(ctypes.byref(batch), int(threads), offsets.ctypes.data, None, 0, ctypes.byref(nbytes), status.ctypes.data))
    data = np.empty(int(nbytes.value), dtype=np.uint8)  # the sizing call returns an upper bound; the molecules are packed once
    check(lib.pmx_pack_features(ctypes.byref(batch), int(threads), offsets.ctypes.data, data.ctypes.data, data.size, ctypes.byref(nbytes),
                                status.ctypes.data))
    return PackedLibrary(offsets, data[: int(nbytes.value)]), status


# A header-only record (0 nodes, 0 conformers): the engine reports PMX_LIGAND_UNSUPPORTED and a NaN score for it.
UNSUPPORTED_RECORD = _HEADER.pack(0, 0, 0, 0) + b"\0" * (RECORD_ALIGN - _HEADER.size)


def pack_ligand_or_marker(lig: LigandFeatures) -> tuple[bytes, str | None]:
    """`pack_ligand`, except that a ligand outside the engine's structural limits becomes `UNSUPPORTED_RECORD`
    (returned with the reason) instead of raising: one oversized molecule must not abort a whole screen."""
    try:
        return pack_ligand(lig), None
    except LigandTooLarge as e:
        return UNSUPPORTED_RECORD, str(e)


@dataclass
class PackedLibrary:
    offsets: np.ndarray  # u64 [N + 1]
    data: np.ndarray  # u8 [offsets[-1]]

    def __len__(self) -> int:
        return int(self.offsets.shape[0]) - 1

    @classmethod
    def from_records(cls, records: Sequence[bytes]) -> "PackedLibrary":
        sizes = np.fromiter((len(r) for r in records), dtype=np.uint64, count=len(records))
        if np.any(sizes % RECORD_ALIGN):
            raise ValueError("records must be padded to 16 bytes")
        offsets = np.zeros(len(records) + 1, dtype=np.uint64)
        np.cumsum(sizes, out=offsets[1:])
        data = np.frombuffer(b"".join(records), dtype=np.uint8).copy()
        return cls(offsets, data)

    def record(self, i: int) -> bytes:
        return self.data[int(self.offsets[i]) : int(self.offsets[i + 1])].tobytes()

    def header(self, i: int) -> tuple[int, int, int]:
        """(n_nodes, n_conf, n_clusters) of ligand i."""
        n, c, k, _ = _HEADER.unpack_from(self.data, int(self.offsets[i]))
        return n, c, k

    def headers(self) -> np.ndarray:
        """u16 [N, 3]: n_nodes, n_conf, n_clusters."""
        starts = self.offsets[:-1].astype(np.int64)
        raw = self.data
        out = np.empty((len(self), 3), dtype=np.uint16)
        for col in range(3):
            lo = raw[starts + 2 * col].astype(np.uint16)
            hi = raw[starts + 2 * col + 1].astype(np.uint16)
            out[:, col] = lo | (hi << 8)
        return out

    def num_conformers(self) -> np.ndarray:
        return self.headers()[:, 1].astype(np.int64)

    def slice(self, first: int, count: int) -> "PackedLibrary":
        lo, hi = int(self.offsets[first]), int(self.offsets[first + count])
        return PackedLibrary(self.offsets[first : first + count + 1] - np.uint64(lo), self.data[lo:hi])

    def unpack(self, i: int) -> dict:
        """Decode record i (for tests and debugging)."""
        n, c, k = self.header(i)
        base = int(self.offsets[i]) + _HEADER.size
        typemask = self.data[base : base + n].copy()
        ends = self.data[base + n : base + n + k].astype(np.int64)
        off = base + n + k
        off += (-(off - int(self.offsets[i]))) % 4
        xyz = self.data[off : off + 12 * n * c].view(np.float32).reshape(n, 3, c).copy()
        return dict(n_nodes=n, n_conf=c, n_clusters=k, typemask=typemask, cluster_end=ends, xyz=xyz)

    # -- storage ---------------------------------------------------------------
    def save(self, path: str | Path) -> None:
        with open(path, "wb") as w:
            w.write(_MAGIC)
            w.write(struct.pack("<QQ", len(self), int(self.data.shape[0])))
            w.write(np.ascontiguousarray(self.offsets, dtype="<u8").tobytes())
            w.write(np.ascontiguousarray(self.data).tobytes())

    @classmethod
    def load(cls, path: str | Path) -> "PackedLibrary":
        with open(path, "rb") as f:
            if f.read(8) != _MAGIC:
                raise ValueError(f"{path}: not a packed ligand library")
            n, nbytes = struct.unpack("<QQ", f.read(16))
            offsets = np.frombuffer(f.read(8 * (n + 1)), dtype="<u8").astype(np.uint64)
            data = np.frombuffer(f.read(nbytes), dtype=np.uint8).copy()
        if offsets.shape[0] != n + 1 or data.shape[0] != nbytes or int(offsets[-1]) != nbytes:
            raise ValueError(f"{path}: truncated library")
        return cls(offsets, data)


def as_packed_library(obj) -> PackedLibrary:
    """Accept the ligand forms the scoring API takes and return a `PackedLibrary`."""
    if isinstance(obj, PackedLibrary):
        return obj
    if isinstance(obj, (bytes, bytearray)):
        return PackedLibrary.from_records([bytes(obj)])
    if isinstance(obj, LigandFeatures):
        return PackedLibrary.from_records([pack_ligand(obj)])
    if isinstance(obj, ClusteredLigand):
        return PackedLibrary.from_records([pack_clustered_ligand(obj)])
    features = getattr(obj, "features", None)  # pharmaconet_amd.ligand.Ligand
    if isinstance(features, LigandFeatures):
        return PackedLibrary.from_records([pack_ligand(features)])
    if isinstance(obj, (list, tuple)):
        records: list[bytes] = []
        for item in obj:
            records.append(as_packed_library(item).record(0))
        return PackedLibrary.from_records(records)
    raise TypeError(f"cannot pack {type(obj).__name__} as a ligand library")
