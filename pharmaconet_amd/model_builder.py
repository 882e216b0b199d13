"""Hotspot density maps -> pharmacophore model state.

This is what `PharmacophoreModel.create` (`src/pmnet/pharmacophore_model.py:108-149`) obtains from
`DensityMapGraph` (`src/pmnet/utils/density_map.py:28-181`), restated as one function that goes from the
hotspot list straight to the state dict of Appendix A (`pharmacophore_model.py:178-190`); there is no
intermediate object graph. Runs on the host, once per pocket; it is not on the screening hot path.

The result is meant to be *equal* to the reference's, including the things that only follow from CPython
container behaviour. Those are kept by doing the same container operations in the same order:

* voxel components are seeded by `set.pop()` on a set of `(x, y, z)` tuples filled in `np.where` order and
  emptied in breadth-first / 26-neighbour order (`density_map.py:91-110`) -> node numbering;
* cluster members are held in sets of node indices (the reference's nodes hash as their index,
  `density_map.py:237-238`) built by the same add / update sequence -> order of the float32 mean and of the
  `node_indices` tuples (`density_map.py:196-203`, `pharmacophore_model.py:231-247`).

Only `node_types` (a tuple made from a set of strings) has no defined order in the reference either (string
hashing is salted per process); it is emitted sorted.
"""

from __future__ import annotations

import math
from typing import Any, Iterable, Sequence

import numpy as np

# data/constant.py:3-14 -- interaction (NCI) types in node_dict order
INTERACTION_TYPES: tuple[str, ...] = (
    "Hydrophobic",
    "PiStacking_P",
    "PiStacking_T",
    "PiCation_lring",
    "PiCation_pring",
    "HBond_ldon",
    "HBond_pdon",
    "SaltBridge_lneg",
    "SaltBridge_pneg",
    "XBond",
)

# pharmacophore_model.py:22-33 -- interaction type of a pocket hotspot -> pharmacophore type asked of the ligand
PHARMACOPHORE_OF: dict[str, str] = {
    "Hydrophobic": "Hydrophobic",
    "PiStacking_P": "Aromatic",
    "PiStacking_T": "Aromatic",
    "PiCation_lring": "Aromatic",
    "PiCation_pring": "Cation",
    "HBond_pdon": "HBond_acceptor",
    "HBond_ldon": "HBond_donor",
    "SaltBridge_pneg": "Cation",
    "SaltBridge_lneg": "Anion",
    "XBond": "Halogen",
}

OVERLAP_DISTANCE = 1.5  # density_map.py:12
CLUSTER_DISTANCE = 3.0  # density_map.py:13
MIN_COMPONENT_VOXELS = 8  # density_map.py:60-61

# density_map.py:122-137: charged / aromatic centres absorb nearby nodes of a minor kind
_GROUP_RULES: tuple[tuple[str, tuple[str, ...], str], ...] = (
    ("Cation", ("SaltBridge_pneg", "PiCation_pring"), "HBond"),
    ("Anion", ("SaltBridge_lneg",), "HBond"),
    ("Aromatic", ("PiStacking", "PiCation_lring"), "Hydrophobic"),
)
# density_map.py:163-167
_SINGLE_RULES: tuple[tuple[str, str], ...] = (("HBond", "HBond"), ("Hydrophobic", "Hydrophobic"), ("Halogen", "XBond"))
# density_map.py:45-47 -- key order of node_cluster_dict
_CLUSTER_KINDS: tuple[str, ...] = ("Cation", "Anion", "HBond", "Aromatic", "Hydrophobic", "Halogen")

_NEIGHBOUR_STEPS: tuple[tuple[int, int, int], ...] = tuple(
    (dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1) if (dx, dy, dz) != (0, 0, 0)
)


def voxel_components(mask: np.ndarray) -> Iterable[tuple[list[tuple[int, int, int]], list[float]]]:
    """26-connected components of `mask > 0` with the voxel values, in the reference's order
    (`density_map.py:79-110`): seeds come from `set.pop()`, members in breadth-first discovery order."""
    xs, ys, zs = np.where(mask > 0.0)
    left = {(int(x), int(y), int(z)) for x, y, z in zip(xs, ys, zs)}
    while left:
        seed = left.pop()
        members = [seed]
        values = [float(mask[seed])]
        head = 0
        while head < len(members):
            x, y, z = members[head]
            head += 1
            for dx, dy, dz in _NEIGHBOUR_STEPS:
                q = (x + dx, y + dy, z + dz)
                if q in left:
                    left.remove(q)
                    members.append(q)
                    values.append(float(mask[q]))
        yield members, values


def voxel_components_device(masks: Sequence[np.ndarray], device: int = 0) -> list[list[tuple[np.ndarray, np.ndarray]]]:
    """`voxel_components` for all hotspot maps of a pocket at once, the searches on the GPU (`csrc/pmx_density.hip`): per map
    the list of (members int64 [n, 3], values float64 [n]) in the reference's order. What stays here is what only CPython can
    say - which voxel `set.pop()` returns next (`density_map.py:91-92`): the set is built as the reference builds it, a popped
    seed's whole component (known from the device's labels) is discarded from it voxel by voxel - discarding marks the same table
    slots whatever the order, so every later `pop()` returns what the reference's returns - and the device then lists every component
    from its seed in breadth-first discovery order (`:93-109`)."""
    import ctypes
    from collections import deque

    from . import _ffi

    lib = _ffi.load()
    size = int(masks[0].shape[0])
    stack = np.ascontiguousarray(np.stack([np.asarray(m, dtype=np.float32) for m in masks]))
    assert stack.shape[1:] == (size, size, size)
    # the device sees float32 maps, the voxel set below the maps as given (density_map.py:91): a float64 value that underflows in
    # float32 would be active here and not there
    for m, f32 in zip(masks, stack):
        if np.asarray(m).dtype != np.float32 and np.any((np.asarray(m) > 0.0) != (f32 > 0.0)):
            raise RuntimeError("voxel_components_device: a map's active voxels change when cast to float32; use the host search (device=None)")
    handle = ctypes.c_void_p()
    _ffi.check(lib.pmx_density_create(stack.ctypes.data, len(masks), size, int(device), ctypes.byref(handle)))
    try:
        labels = np.empty(stack.size, dtype=np.int32)
        _ffi.check(lib.pmx_density_labels(handle, labels.ctypes.data))
        labels = labels.reshape(len(masks), -1)
        comp_map: list[int] = []
        comp_seed: list[int] = []
        comp_size: list[int] = []
        for mi, mask in enumerate(masks):
            xs, ys, zs = np.where(mask > 0.0)
            left = {(int(x), int(y), int(z)) for x, y, z in zip(xs, ys, zs)}  # as density_map.py:91-92 builds it
            lin = (xs.astype(np.int64) * size + ys) * size + zs
            lab = labels[mi][lin]
            order = np.argsort(lab, kind="stable")
            bounds = np.flatnonzero(np.diff(lab[order], prepend=-2))
            groups = {int(lab[order[b]]): order[b:e] for b, e in zip(bounds, list(bounds[1:]) + [len(order)])}
            while left:
                seed = left.pop()
                sl = (seed[0] * size + seed[1]) * size + seed[2]
                idx = groups[int(labels[mi][sl])]
                # (one discard per voxel, as the reference's `remove` calls: `difference_update` rebuilds the table once a
                # quarter of it is dummies, which would change what pop() returns next)
                deque(map(left.discard, zip(xs[idx].tolist(), ys[idx].tolist(), zs[idx].tolist())), maxlen=0)
                comp_map.append(mi)
                comp_seed.append(sl)
                comp_size.append(len(idx))
        n = len(comp_map)
        offsets = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(comp_size, out=offsets[1:])
        members = np.empty(int(offsets[-1]), dtype=np.int32)
        cm, cs = np.asarray(comp_map, dtype=np.int32), np.asarray(comp_seed, dtype=np.int32)
        _ffi.check(lib.pmx_density_order(handle, n, cm.ctypes.data, cs.ctypes.data, offsets.ctypes.data, members.ctypes.data))
    finally:
        lib.pmx_density_destroy(handle)
    if members.size and members.min() < 0:
        raise RuntimeError("pmx_density_order left a component incomplete")
    out: list[list[tuple[np.ndarray, np.ndarray]]] = [[] for _ in masks]
    for c in range(n):
        lin = members[offsets[c]:offsets[c + 1]].astype(np.int64)
        coords = np.stack([lin // (size * size), (lin // size) % size, lin % size], axis=1)
        vals = np.asarray(masks[comp_map[c]], dtype=np.float32).reshape(-1)[lin].astype(np.float64)
        out[comp_map[c]].append((coords, vals))
    return out


_DEVICE_SEARCH_OK: dict[int, bool] = {}


def device_search_agrees(device: int) -> bool:
    """Once per process and device: does `voxel_components_device` list the components of a synthetic map (four blobs, some 3 000 active
    voxels: large enough for the interpreter's set to resize and collect dummies) in the order the reference's own loop
    (`voxel_components`) does? Discarding a popped seed's component voxel by voxel gives every later `pop()` what the reference's gives
    on CPython 3.8-3.12 as tested; on an interpreter whose `set` behaves differently the model would still be a valid model, but not
    the reference's state - so the device search is used only where this check passes (`build_model_state` falls back to the host loop)."""
    if device not in _DEVICE_SEARCH_OK:
        rng = np.random.default_rng(20250930)
        size = 32
        grid = np.indices((size, size, size)).astype(np.float32)
        mask = np.zeros((size, size, size), dtype=np.float32)
        for centre, radius in (((8, 8, 8), 5.2), ((22, 9, 20), 5.8), ((10, 23, 21), 4.6), ((24, 24, 6), 3.3)):
            d2 = sum((grid[k] - centre[k]) ** 2 for k in range(3))
            mask[d2 < radius * radius] = 1.0
        mask *= rng.uniform(0.05, 1.0, size=mask.shape).astype(np.float32)
        # (a device that is not there, or a libpmx that does not load, raises from here as it would from the build itself)
        host = [(np.asarray(m, dtype=np.int64), np.asarray(v, dtype=np.float64)) for m, v in voxel_components(mask)]
        dev = voxel_components_device([mask], device)[0]
        ok = len(host) == len(dev) and all(np.array_equal(hm, dm) and np.array_equal(hv, dv) for (hm, hv), (dm, dv) in zip(host, dev))
        _DEVICE_SEARCH_OK[device] = ok
        if not ok:
            import warnings

            warnings.warn("pharmaconet_amd: the device voxel search does not reproduce this interpreter's set.pop() order; "
                          "PharmacophoreModel.create uses the host search")
    return _DEVICE_SEARCH_OK[device]


def _grid_to_world(coords, center, resolution: float, size: int) -> tuple[float, float, float]:
    """density_map.py:16-25: voxel coordinates -> Angstrom, the box being centred on `center`."""
    half = resolution * (size - 1) / 2
    return tuple(float((c - half) + g * resolution) for c, g in zip(center, coords))  # type: ignore[return-value]


def build_model_state(
    pdbblock: str | None,
    center: Sequence[float] | np.ndarray,
    hotspot_infos: Sequence[dict],
    resolution: float = 0.5,
    size: int = 64,
    device: int | None = None,
) -> dict[str, Any]:
    """State dict (`.pm` / `.json` schema) of the model that `PharmacophoreModel.create` builds from
    `hotspot_infos` = [{nci_type, hotspot_position, hotspot_score, point_map}] (`pharmacophore_model.py:108-131`).
    `device`: GPU ordinal - the voxel searches of all hotspots run there (`voxel_components_device`); None: on the host."""
    assert len(center) == 3
    if not isinstance(center, tuple):
        center = tuple(np.asarray(center).tolist())

    # ---- nodes and edges (density_map.py:49-74, 206-278); an edge joins every node pair and each node to itself
    centers: list[np.ndarray] = []  # float32[3]
    radii: list[float] = []
    kinds: list[str] = []
    nodes: list[dict[str, Any]] = []
    edges: list[dict[str, Any]] = []
    mean_of: dict[tuple[int, int], float] = {}
    if device is not None and not device_search_agrees(device):
        device = None
    searched = voxel_components_device([info["point_map"] for info in hotspot_infos], device) if device is not None and len(hotspot_infos) else None
    for h, info in enumerate(hotspot_infos):
        kind = info["nci_type"]
        hx, hy, hz = tuple(np.asarray(info["hotspot_position"]).tolist())
        score = float(info["hotspot_score"])
        for members, values in (searched[h] if searched is not None else voxel_components(info["point_map"])):
            if len(members) < MIN_COMPONENT_VOXELS:
                continue
            grids = np.array(members)
            centroid = np.average(grids, axis=0, weights=np.array(values))  # density-weighted, in voxel units
            me = len(nodes)
            centers.append(np.array(_grid_to_world(centroid, center, resolution, size), dtype=np.float32))
            radii.append((grids.shape[0] / (4 * math.pi / 3)) ** (1 / 3) * resolution)  # sphere of equal volume
            kinds.append(kind)
            nodes.append(
                dict(
                    index=me,
                    type=PHARMACOPHORE_OF[kind],
                    interaction_type=kind,
                    hotspot_position=(hx, hy, hz),
                    score=score,
                    center=tuple(centers[me].tolist()),
                    radius=radii[me],
                    neighbor_edge_dict={},
                    overlapped_nodes=[],
                )
            )
            for other in range(me + 1):  # earlier nodes, then the self loop
                e = len(edges)
                lo, hi = (other, me)
                dist = np.linalg.norm(centers[lo] - centers[hi]).item()
                edges.append(
                    dict(
                        index=e,
                        node_indices=(lo, hi),
                        edge_type=(min(kinds[lo], kinds[hi]), max(kinds[lo], kinds[hi])),
                        distance_mean=dist,
                        distance_std=math.sqrt(radii[lo] ** 2 + radii[hi] ** 2),
                    )
                )
                mean_of[(lo, hi)] = mean_of[(hi, lo)] = dist
                nodes[me]["neighbor_edge_dict"][other] = e
                nodes[other]["neighbor_edge_dict"][me] = e
                if dist < OVERLAP_DISTANCE:
                    nodes[me]["overlapped_nodes"].append(other)
                    nodes[other]["overlapped_nodes"].append(me)

    n = len(nodes)

    def close(i: int, j: int) -> bool:
        return mean_of[(i, j)] < CLUSTER_DISTANCE

    # ---- clusters (density_map.py:112-181)
    clusters: dict[str, list[dict[str, Any]]] = {k: [] for k in _CLUSTER_KINDS}
    used: set[int] = set()

    def emit(kind: str, members: set[int]) -> None:
        order = list(members)  # set iteration order: the order of the reference's float32 mean
        pos = np.array([centers[i] for i in order])
        reach = np.array([radii[i] * 2 for i in order])
        mid = np.mean(pos, axis=0)
        extent = np.linalg.norm(pos - mid.reshape(1, 3), axis=-1) + reach
        cx, cy, cz = mid.tolist()
        clusters[kind].append(
            dict(
                cluster_type=kind,
                node_indices=tuple(set({i for i in members})),
                node_types=tuple(sorted({PHARMACOPHORE_OF[kinds[i]] for i in members})),
                center=(cx, cy, cz),
                size=np.max(extent).item(),
            )
        )

    for i in range(n):
        if i in used:
            continue
        for kind, major, minor in _GROUP_RULES:
            if not kinds[i].startswith(major):
                continue
            members = {i}
            members.update(j for j in nodes[i]["overlapped_nodes"] if kinds[j].startswith(major))
            for j in range(n):  # a minor node joins if it is close to any member so far, earlier minors included
                if kinds[j].startswith(minor) and any(close(j, m) for m in members):
                    members.add(j)
            used.update(members)
            emit(kind, members)
            break
    for i in range(n):
        if i in used:
            continue
        for kind, prefix in _SINGLE_RULES:
            if not kinds[i].startswith(prefix):
                continue
            members = {j for j in range(n) if kinds[j].startswith(prefix) and close(i, j)}
            members.add(i)
            emit(kind, members)
            used.update(members)
            break

    by_kind: dict[str, list[int]] = {k: [] for k in INTERACTION_TYPES}
    for i, kind in enumerate(kinds):
        by_kind[kind].append(i)
    return dict(pdbblock=pdbblock, nodes=nodes, edges=edges, node_cluster_dict=clusters, node_dict=by_kind)
