"""`PharmacophoreModel`: the `.pm` / `.json` model object and the scoring entry points.

Host-side mirror of the reference's `src/pmnet/pharmacophore_model.py`:

* `load` / `save` and the state schema follow `pharmacophore_model.py:151-204`
  (`.pm` = pickle of a builtins-only dict, `.json` = the same dict; any other
  extension raises `NotImplementedError`, `:160-161,172-173`).
* `scoring_file` / `scoring_smiles` / `scoring_pbmol` / `_scoring` keep the
  reference's names and argument meaning (`pharmacophore_model.py:60-106`).
  They return a Python float computed by the HIP engine (`pharmaconet_amd.engine`);
  there is no CPU scoring path in this package.
* `screen` is the batched form of `screening.py:46-75`: one packed library in,
  per-ligand scores (and optionally a top-k) out.

What the device consumes is `FlatModel`: dense `[Nm, Nm]` float32 edge tables
(the `model_node1.neighbor_edge_dict[model_node2]` lookup of
`match_utils.py:35-48` made positional), node types, and the cluster list in
`model.node_clusters` order (`pharmacophore_model.py:202-204`).
"""

from __future__ import annotations

import io
import json
import math
import os
import pickle
from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np

from .constants import MAX_MODEL_CLUSTERS, MAX_MODEL_NODES, TYPE_ID

__all__ = ["PharmacophoreModel", "FlatModel"]


class _BuiltinsOnlyUnpickler(pickle.Unpickler):
    """A `.pm` file holds dict/list/tuple/str/int/float only (`pharmacophore_model.py:178-189`)."""

    def find_class(self, module: str, name: str):  # pragma: no cover - defensive
        raise pickle.UnpicklingError(f".pm files contain builtins only; refusing to import {module}.{name}")


@dataclass
class FlatModel:
    """Positional tables of one pharmacophore model (host arrays handed to `pmx_model_create`)."""

    node_type: np.ndarray  # u8 [Nm]       type id of each model node
    edge_mean: np.ndarray  # f32 [Nm, Nm]  distance_mean of edge (m, n), symmetric, self-loops on the diagonal
    edge_std: np.ndarray  # f32 [Nm, Nm]   distance_std
    cluster_nodes: np.ndarray  # u64 [K] (models of up to 64 nodes) or [K, ceil(Nm / 64)]: bit m % 64 of word m // 64 set <=> node m in cluster
    cluster_typemask: np.ndarray  # u8 [K] bit t set <=> type t in the cluster's stored node_types
    cluster_center: np.ndarray  # f64 [K, 3]
    cluster_size: np.ndarray  # f64 [K]
    cluster_type: tuple[str, ...]

    @property
    def num_nodes(self) -> int:
        return int(self.node_type.shape[0])

    @property
    def num_clusters(self) -> int:
        return int(self.cluster_nodes.shape[0])


def _node_masks(masks: list[int], nm: int) -> np.ndarray:
    """Python-int node sets -> the words `pmx_model_desc.cluster_nodes` holds (one per cluster up to 64 nodes)."""
    words = max(1, (nm + 63) // 64)
    out = np.array([[(m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(words)] for m in masks], dtype=np.uint64).reshape(len(masks), words)
    return out[:, 0].copy() if words == 1 else out


def cluster_node_sets(flat: "FlatModel") -> list[int]:
    """The node set of every cluster as a Python int (bit m <=> node m), whatever the word count."""
    cn = np.asarray(flat.cluster_nodes, dtype=np.uint64).reshape(flat.num_clusters, -1)
    return [sum(int(cn[k, w]) << (64 * w) for w in range(cn.shape[1])) for k in range(cn.shape[0])]


def _flatten_state(state: dict[str, Any]) -> FlatModel:
    nodes = state["nodes"]
    edges = state["edges"]
    nm = len(nodes)
    if nm > MAX_MODEL_NODES:
        raise ValueError(f"model has {nm} nodes; this implementation supports at most {MAX_MODEL_NODES}")
    for pos, node in enumerate(nodes):
        if int(node["index"]) != pos:
            raise ValueError("model node indices must be positional (pharmacophore_model.py:193,218)")
    node_type = np.array([TYPE_ID[node["type"]] for node in nodes], dtype=np.uint8)

    mean64 = np.full((nm, nm), np.nan, dtype=np.float64)
    std64 = np.full((nm, nm), np.nan, dtype=np.float64)
    for node in nodes:
        i = int(node["index"])
        # json stores the neighbour index as str (pharmacophore_model.py:279-281)
        for nbr, edge_index in node["neighbor_edge_dict"].items():
            edge = edges[int(edge_index)]
            mean64[i, int(nbr)] = float(edge["distance_mean"])
            std64[i, int(nbr)] = float(edge["distance_std"])
    # match_utils.py:35-48 builds float32 arrays of the means / stds
    edge_mean = mean64.astype(np.float32)
    edge_std = std64.astype(np.float32)

    cluster_nodes: list[int] = []
    cluster_typemask: list[int] = []
    centers: list[tuple[float, float, float]] = []
    sizes: list[float] = []
    ctypes_: list[str] = []
    # model.node_clusters = concatenation of node_cluster_dict.values() (pharmacophore_model.py:202-204)
    for cluster_list in state["node_cluster_dict"].values():
        for cluster in cluster_list:
            mask = 0
            for index in cluster["node_indices"]:
                mask |= 1 << int(index)
            tmask = 0
            for typ in cluster["node_types"]:
                tmask |= 1 << TYPE_ID[typ]
            cluster_nodes.append(mask)
            cluster_typemask.append(tmask)
            x, y, z = cluster["center"]
            centers.append((float(x), float(y), float(z)))
            sizes.append(float(cluster["size"]))
            ctypes_.append(str(cluster["cluster_type"]))
    if len(cluster_nodes) > MAX_MODEL_CLUSTERS:
        raise ValueError(
            f"model has {len(cluster_nodes)} clusters; this implementation supports at most {MAX_MODEL_CLUSTERS}"
        )
    # every (m, n) the matcher can look up must have an edge: m in cluster a, n in cluster b, any a, b
    in_cluster = 0
    for mask in cluster_nodes:
        in_cluster |= mask
    used = [m for m in range(nm) if (in_cluster >> m) & 1]
    for m in used:
        for n in used:
            if math.isnan(mean64[m, n]):
                raise ValueError(f"model edge ({m}, {n}) is missing from neighbor_edge_dict")
    edge_mean = np.nan_to_num(edge_mean, nan=0.0)
    edge_std = np.nan_to_num(edge_std, nan=1.0)

    return FlatModel(
        node_type=node_type,
        edge_mean=np.ascontiguousarray(edge_mean),
        edge_std=np.ascontiguousarray(edge_std),
        cluster_nodes=_node_masks(cluster_nodes, nm),
        cluster_typemask=np.array(cluster_typemask, dtype=np.uint8),
        cluster_center=np.array(centers, dtype=np.float64).reshape(-1, 3),
        cluster_size=np.array(sizes, dtype=np.float64),
        cluster_type=tuple(ctypes_),
    )


class PharmacophoreModel:
    """Pickle-friendly pharmacophore model (`pharmacophore_model.py:50-58`)."""

    def __init__(self):
        self._state: dict[str, Any] | None = None
        self._flat: FlatModel | None = None
        self._engine_handle = None  # device tables, created on first scoring call, never pickled

    # ------------------------------------------------------------------ I/O
    @classmethod
    def load(cls, save_path: str | Path) -> "PharmacophoreModel":
        extension = os.path.splitext(save_path)[-1]
        if extension == ".pm":
            with open(save_path, "rb") as f:
                state = _BuiltinsOnlyUnpickler(io.BytesIO(f.read())).load()
        elif extension == ".json":
            with open(save_path) as f:
                state = json.load(f)
        else:
            raise NotImplementedError
        model = cls()
        model.__setstate__(state)
        return model

    def save(self, save_path: str | Path) -> None:
        extension = os.path.splitext(save_path)[-1]
        state = self.__getstate__()
        if extension == ".pm":
            with open(save_path, "wb") as w:
                pickle.dump(state, w)
        elif extension == ".json":
            with open(save_path, "w") as w:
                json.dump(state, w, indent=2)
        else:
            raise NotImplementedError

    def __getstate__(self) -> dict[str, Any]:
        assert self._state is not None, "empty model"
        return self._state

    def __setstate__(self, state: dict[str, Any]) -> None:
        self._state = state
        self._flat = _flatten_state(state)
        self._engine_handle = None
        self._object_graph = None

    @classmethod
    def create(cls, pdbblock, center, hotspot_infos, resolution: float = 0.5, size: int = 64, device="auto"):
        """Hotspot density maps -> model (`pharmacophore_model.py:108-149`): same arguments as the reference;
        `hotspot_infos` = [{nci_type, hotspot_position, hotspot_score, point_map}]. The graph build of
        `utils/density_map.py` is restated in `pharmaconet_amd.model_builder`: the voxel searches of all hotspots run on the GPU
        (`device`: ordinal, "auto" = the current GPU if one is visible, None = the host loop), the rest - a few dozen nodes - on
        the host; the state is the reference's either way."""
        from .model_builder import build_model_state

        model = cls()
        auto = device == "auto"
        if auto:
            device = None
            try:
                import torch

                if torch.cuda.is_available():
                    device = torch.cuda.current_device()
            except Exception:
                device = None
        try:
            state = build_model_state(pdbblock, center, hotspot_infos, resolution, size, device=device)
        except (RuntimeError, OSError):  # (_ffi.PmxError is a RuntimeError)
            # "auto" promises a model, not a GPU: libpmx.so missing or unloadable, or maps the device search cannot take
            # (model_builder.voxel_components_device) - the host search gives the same state
            if not auto or device is None:
                raise
            state = build_model_state(pdbblock, center, hotspot_infos, resolution, size, device=None)
        model.__setstate__(state)
        return model

    # ------------------------------------------------------------ accessors
    @property
    def pdbblock(self) -> str | None:
        assert self._state is not None
        return self._state.get("pdbblock")

    @property
    def flat(self) -> FlatModel:
        assert self._flat is not None, "empty model"
        return self._flat

    @property
    def num_nodes(self) -> int:
        return self.flat.num_nodes

    # The reference's object graph (`pharmacophore_model.py:191-204,207-365`), built on first use from the state dict: code
    # written against `model.nodes / .edges / .node_dict / .node_cluster_dict / .node_clusters` keeps working. Read-only views -
    # scoring never touches them (the device tables come from `flat`).
    def _graph(self):
        g = getattr(self, "_object_graph", None)
        if g is None:
            assert self._state is not None, "empty model"
            st = self._state
            nodes = [ModelNode(self, **kw) for kw in st["nodes"]]
            g = dict(nodes=nodes)
            self._object_graph = g  # (ModelEdge / ModelNodeCluster look the nodes up through the model)
            g["edges"] = [ModelEdge(self, **kw) for kw in st["edges"]]
            for node in nodes:
                node.setup()
            g["node_dict"] = {typ: [nodes[i] for i in indices] for typ, indices in st["node_dict"].items()}
            g["node_cluster_dict"] = {typ: [ModelNodeCluster(self, **kw) for kw in lst] for typ, lst in st["node_cluster_dict"].items()}
            g["node_clusters"] = [c for lst in g["node_cluster_dict"].values() for c in lst]
        return g

    @property
    def nodes(self) -> list["ModelNode"]:
        return self._graph()["nodes"]

    @property
    def edges(self) -> list["ModelEdge"]:
        return self._graph()["edges"]

    @property
    def node_dict(self) -> dict[str, list["ModelNode"]]:
        return self._graph()["node_dict"]

    @property
    def node_cluster_dict(self) -> dict[str, list["ModelNodeCluster"]]:
        return self._graph()["node_cluster_dict"]

    @property
    def node_clusters(self) -> list["ModelNodeCluster"]:
        return self._graph()["node_clusters"]

    @property
    def num_clusters(self) -> int:
        return self.flat.num_clusters

    # -------------------------------------------------------------- scoring
    def scoring_file(
        self,
        ligand_file: str | Path,
        weights: dict[str, float] | None = None,
        num_conformers: int | None = None,
    ) -> float:
        """`pharmacophore_model.py:83-90`."""
        from .ligand import Ligand

        ligand = Ligand.load_from_file(ligand_file, num_conformers)
        return self._scoring(ligand, weights)

    def scoring_smiles(
        self,
        ligand_smiles: str,
        num_conformers: int,
        weights: dict[str, float] | None = None,
    ) -> float:
        """`pharmacophore_model.py:92-99`."""
        from .ligand import Ligand

        ligand = Ligand.load_from_smiles(ligand_smiles, num_conformers)
        return self._scoring(ligand, weights)

    def scoring_pbmol(
        self,
        ligand_pbmol,
        atom_positions,
        conformer_axis: int | None = None,
        weights: dict[str, float] | None = None,
    ) -> float:
        """`pharmacophore_model.py:60-81`."""
        from .ligand import Ligand

        ligand = Ligand(ligand_pbmol, atom_positions, conformer_axis)
        return self._scoring(ligand, weights)

    def _scoring(self, ligand, weights: dict[str, float] | None = None) -> float:
        """`pharmacophore_model.py:101-106`: GraphMatcher(self, ligand, weights).run(), on the GPU.

        `ligand` is anything `pharmaconet_amd.library.as_packed_library` accepts: a `Ligand`,
        a `LigandFeatures`, packed record bytes or a one-ligand `PackedLibrary`.
        """
        from .engine import score_one

        return score_one(self, ligand, weights)

    def screen(self, library, weights: dict[str, float] | None = None, topk: int | None = None, **kwargs):
        """Batched `screening.py:46-75`: score every ligand of a packed library on the GPU.

        Returns a `ScreeningResult` (per-ligand float32 scores in library order; with `topk`,
        also the k best `(index, score)` in the order `screening.py:70` would list them)."""
        from .engine import screen

        return screen(self, library, weights=weights, topk=topk, **kwargs)


class ModelNodeCluster:
    """`pharmacophore_model.py:207-246` (a view; `get_kwargs()` gives back the state entry)."""

    def __init__(self, graph, cluster_type, node_indices, node_types, center, size):
        self.type = cluster_type
        self.nodes = {graph.nodes[int(i)] for i in node_indices}
        self.node_indices = {int(i) for i in node_indices}
        self.node_types = set(node_types)
        self.center = tuple(center)
        self.size = size

    def __repr__(self):
        return f"ModelCluster({self.type})[{self.nodes}]"

    def get_kwargs(self):
        return dict(cluster_type=self.type, node_indices=tuple(self.node_indices), node_types=tuple(self.node_types), center=self.center, size=self.size)


class ModelNode:
    """`pharmacophore_model.py:249-320`."""

    def __init__(self, graph, index, type, interaction_type, hotspot_position, score, center, radius, neighbor_edge_dict, overlapped_nodes):
        self.graph = graph
        self.index = int(index)
        self.type = type
        self.interaction_type = interaction_type
        self.hotspot_position = tuple(hotspot_position)
        self.score = score
        self.center = tuple(center)
        self.radius = radius
        self._neighbor_edge_dict = {int(k): int(v) for k, v in neighbor_edge_dict.items()}  # (.json keys are strings, :279)
        self._overlapped_nodes = [int(i) for i in overlapped_nodes]
        self.neighbor_edge_dict = {}
        self.overlapped_nodes = []

    def setup(self):
        self.neighbor_edge_dict = {self.graph.nodes[n]: self.graph.edges[e] for n, e in self._neighbor_edge_dict.items()}
        self.overlapped_nodes = [self.graph.nodes[i] for i in self._overlapped_nodes]

    def __hash__(self):
        return self.index

    def __eq__(self, other):
        return self is other

    def __repr__(self):
        return f"ModelNode({self.index})[{self.interaction_type}]"

    def get_kwargs(self):
        return dict(index=self.index, type=self.type, interaction_type=self.interaction_type, hotspot_position=self.hotspot_position,
                    score=self.score, center=self.center, radius=self.radius, neighbor_edge_dict=dict(self._neighbor_edge_dict),
                    overlapped_nodes=list(self._overlapped_nodes))


class ModelEdge:
    """`pharmacophore_model.py:323-365`."""

    def __init__(self, graph, index, node_indices, edge_type, distance_mean, distance_std):
        self.graph = graph
        self.index = int(index)
        self.node_indices = (int(node_indices[0]), int(node_indices[1]))
        self.nodes = (graph.nodes[self.node_indices[0]], graph.nodes[self.node_indices[1]])
        self.type = tuple(edge_type)
        self.distance_mean = distance_mean
        self.distance_std = distance_std

    def __hash__(self):
        return self.index

    def __eq__(self, other):
        return self is other

    def __repr__(self):
        return f"ModelEdge({self.node_indices[0]},{self.node_indices[1]})[{self.type[0]},{self.type[1]}]"

    def get_kwargs(self):
        return dict(index=self.index, node_indices=self.node_indices, edge_type=self.type, distance_mean=self.distance_mean, distance_std=self.distance_std)
