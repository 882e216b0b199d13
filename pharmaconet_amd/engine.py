"""Host orchestration of the HIP engine: device handles, batched scoring, top-k.

PyTorch is used for what it is good at here - device buffers, the current HIP stream and (in
`distributed.py`) the RCCL process group; the arithmetic is all in libpmx.so.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .constants import weights_vector
from .library import PackedLibrary, as_packed_library

__all__ = ["DeviceLibrary", "ScreeningResult", "score_one", "screen", "topk", "device_model", "last_score_stats"]


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise _ffi.PmxError("no GPU visible: pharmaconet_amd scores on an MI355X only (there is no CPU path)")
    return torch


def _device_index(device) -> int:
    torch = _torch()
    if device is None:
        return torch.cuda.current_device()
    dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    return dev.index if dev.index is not None else torch.cuda.current_device()


class _ModelHandle:
    def __init__(self, flat, device: int):
        lib = _ffi.load()
        self.device = device
        self._keep = dict(
            node_type=np.ascontiguousarray(flat.node_type, dtype=np.uint8),
            edge_mean=np.ascontiguousarray(flat.edge_mean, dtype=np.float32),
            edge_std=np.ascontiguousarray(flat.edge_std, dtype=np.float32),
            cluster_nodes=np.ascontiguousarray(flat.cluster_nodes, dtype=np.uint64),
            cluster_typemask=np.ascontiguousarray(flat.cluster_typemask, dtype=np.uint8),
            cluster_center=np.ascontiguousarray(flat.cluster_center, dtype=np.float64),
            cluster_size=np.ascontiguousarray(flat.cluster_size, dtype=np.float64),
        )
        desc = _ffi.ModelDesc(
            flat.num_nodes,
            flat.num_clusters,
            *(self._keep[k].ctypes.data for k in (
                "node_type", "edge_mean", "edge_std", "cluster_nodes", "cluster_typemask", "cluster_center", "cluster_size")),
        )
        handle = ctypes.c_void_p()
        _ffi.check(lib.pmx_model_create(ctypes.byref(desc), device, ctypes.byref(handle)))
        self.handle = handle

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _ffi.load().pmx_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def device_model(model, device=None) -> _ModelHandle:
    """Device tables of a `PharmacophoreModel`, created once per (model, device)."""
    dev = _device_index(device)
    cache = model._engine_handle
    if not isinstance(cache, dict):
        cache = {}
        model._engine_handle = cache
    if dev not in cache:
        cache[dev] = _ModelHandle(model.flat, dev)
    return cache[dev]


class DeviceLibrary:
    """A packed ligand library resident in HBM (upload once, score against many models / weights)."""

    def __init__(self, library: PackedLibrary, device=None):
        lib = _ffi.load()
        self.device = _device_index(device)
        offsets = np.ascontiguousarray(library.offsets, dtype=np.uint64)
        data = np.ascontiguousarray(library.data, dtype=np.uint8)
        view = _ffi.LibraryView(len(library), offsets.ctypes.data, data.ctypes.data if data.size else None, 0)
        handle = ctypes.c_void_p()
        _ffi.check(lib.pmx_library_upload(ctypes.byref(view), self.device, ctypes.byref(handle)))
        self.handle = handle
        info = _ffi.LibraryInfo()
        _ffi.check(lib.pmx_library_info_get(self.handle, ctypes.byref(info)))
        self.num_ligands = int(info.n_ligands)
        self.num_bytes = int(info.n_bytes)
        self.total_conformers = int(info.total_conformers)
        self.max_nodes = int(info.max_nodes)
        self.max_conformers = int(info.max_conformers)
        self.max_clusters = int(info.max_clusters)
        self.num_unsupported = int(info.n_unsupported)

    @classmethod
    def from_device_buffers(cls, offsets, data, device=None, adopt: bool = False) -> "DeviceLibrary":
        """A library already in device memory: `offsets` int64/uint64 [N + 1] and `data` uint8 torch tensors - copied, or with `adopt` used in
        place (the tensors are kept alive by the object and must not be written to while it lives)."""
        torch = _torch()
        lib = _ffi.load()
        self = cls.__new__(cls)
        self.device = _device_index(device if device is not None else offsets.device)
        torch.cuda.current_stream(torch.device("cuda", self.device)).synchronize()  # (pmx_library_upload reads a complete view, on the default stream)
        adopt = adopt and data.numel() > 0  # (an empty library has nothing to adopt: torch hands out a null pointer for it)
        if adopt:
            if not (offsets.is_contiguous() and data.is_contiguous()):
                raise ValueError("adopted buffers must be contiguous")
            self._adopted = (offsets, data)
        view = _ffi.LibraryView(int(offsets.numel()) - 1, offsets.data_ptr(), data.data_ptr(), 2 if adopt else 1)
        handle = ctypes.c_void_p()
        _ffi.check(lib.pmx_library_upload(ctypes.byref(view), self.device, ctypes.byref(handle)))
        self.handle = handle
        info = _ffi.LibraryInfo()
        _ffi.check(lib.pmx_library_info_get(self.handle, ctypes.byref(info)))
        self.num_ligands = int(info.n_ligands)
        self.num_bytes = int(info.n_bytes)
        self.total_conformers = int(info.total_conformers)
        self.max_nodes = int(info.max_nodes)
        self.max_conformers = int(info.max_conformers)
        self.max_clusters = int(info.max_clusters)
        self.num_unsupported = int(info.n_unsupported)
        return self

    @classmethod
    def from_features(cls, flat, device=None, check: bool = True) -> "DeviceLibrary":
        """Typed features -> resident library without the host packer: `pack_features_device` + adoption of its buffers.
        `flat`: `library.flatten_features(...)` arrays (NumPy: uploaded; or torch tensors already on the device). With `check`
        a molecule outside the device builder's fixed scratch (status 3) raises - pack such a batch with `pack_features_native`."""
        offsets, data, status = pack_features_device(flat, device)
        if check and bool((status == 3).any()):
            raise _ffi.PmxError("a molecule exceeds the device packer's fixed scratch (include/pmx.h): use library.pack_features_native for this batch")
        self = cls.from_device_buffers(offsets, data, device if device is not None else offsets.device, adopt=True)
        self.pack_status = status
        return self

    def __len__(self) -> int:
        return self.num_ligands

    def close(self) -> None:
        if getattr(self, "handle", None):
            _ffi.load().pmx_library_destroy(self.handle)
            self.handle = None
        self._adopted = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


FEATURE_FIELDS = ("atom_off", "atomic_num", "nbr_off", "nbr", "feat_off", "feat_type", "feat_flags", "feat_atom_off", "feat_atoms",
                  "feat_center_off", "feat_centers", "n_conf", "pos_off", "positions")


def features_to_device(flat, device=None, non_blocking: bool = False) -> dict:
    """The arrays of `library.flatten_features` as device tensors (uint64 offsets travel as int64 bits)."""
    torch = _torch()
    dev = torch.device("cuda", _device_index(device))
    out = {}
    for k in FEATURE_FIELDS:
        a = flat[k]
        if not isinstance(a, torch.Tensor):
            a = np.ascontiguousarray(a)
            a = torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a)
        out[k] = a.to(dev, non_blocking=non_blocking)
    return out


def pack_bound(flat) -> int:
    """Upper bound of the packed size of a feature batch (every feature a node), as `pmx_pack_features`' sizing call computes it."""
    torch = _torch()
    feat_off, n_conf = flat["feat_off"], flat["n_conf"]
    if isinstance(feat_off, torch.Tensor):
        nf = (feat_off[1:] - feat_off[:-1]).to(torch.int64)
        c = n_conf.to(torch.int64).clamp(min=1)
        return int(((((8 + 2 * nf + 3 + 12 * nf * c + 15) // 16) * 16) + 16).sum().item())
    nf = np.diff(np.asarray(feat_off).astype(np.int64))
    c = np.maximum(np.asarray(n_conf).astype(np.int64), 1)
    return int((((8 + 2 * nf + 3 + 12 * nf * c + 15) & ~15) + 16).sum())


def pack_features_device(flat, device=None, out=None, bound: int | None = None):
    """`pmx_pack_features_device` (csrc/pmx_pack_device.hip): LigandGraph + priority sort on the device, on torch's current stream.
    Returns (offsets int64 [n + 1], data uint8 [bytes], status int32 [n]) as device tensors; records byte-identical to
    `library.pack_features_native`. `out` = (offsets, data, status) tensors to write into (data at least `pack_bound(flat)` long)."""
    torch = _torch()
    lib = _ffi.load()
    dev_index = _device_index(device if device is not None else (flat["positions"].device if isinstance(flat["positions"], torch.Tensor) else None))
    tdev = torch.device("cuda", dev_index)
    if not all(isinstance(flat[k], torch.Tensor) and flat[k].is_cuda for k in FEATURE_FIELDS):
        flat = features_to_device(flat, tdev)
    n = int(flat["atom_off"].numel()) - 1
    if out is None:
        cap = int(bound) if bound is not None else pack_bound(flat)
        out = (torch.empty(n + 1, dtype=torch.int64, device=tdev), torch.empty(max(cap, 16), dtype=torch.uint8, device=tdev),
               torch.empty(max(n, 1), dtype=torch.int32, device=tdev))
    offsets, data, status = out
    batch = _ffi.FeatureBatch(n, *(flat[k].data_ptr() for k in FEATURE_FIELDS))
    nbytes = ctypes.c_uint64(0)
    with torch.cuda.device(tdev):
        stream = torch.cuda.current_stream(tdev).cuda_stream
        _ffi.check(lib.pmx_pack_features_device(ctypes.byref(batch), dev_index, ctypes.c_void_p(stream), offsets.data_ptr(), data.data_ptr(), int(data.numel()),
                                                ctypes.byref(nbytes), status.data_ptr()))
    return offsets[: n + 1], data[: int(nbytes.value)], status[:n]


@dataclass
class ScreeningResult:
    scores: "object"  # torch.float32 (float64 with `screen(..., float64=True)`) [count] on the device, library order
    status: "object"  # torch.int32 [count]
    first: int
    topk_scores: "object | None" = None  # torch.float32 [k]
    topk_indices: "object | None" = None  # torch.int64 [k], global ligand indices (first + position)

    def scores_numpy(self) -> np.ndarray:
        return self.scores.cpu().numpy()

    def ranking(self) -> list[tuple[int, float]]:
        """[(ligand index, score)] of the top-k, best first (the order of screening.py:70-75)."""
        assert self.topk_scores is not None and self.topk_indices is not None
        idx = self.topk_indices.cpu().numpy()
        sc = self.topk_scores.cpu().numpy()
        return [(int(i), float(s)) for i, s in zip(idx, sc) if i >= 0]


def _weights_array(weights):
    return (ctypes.c_float * _ffi.NUM_TYPES)(*weights_vector(weights))


def topk(scores, k: int, base_index: int = 0, indices=None):
    """k best of a device float32 tensor: (scores [k], int64 global indices [k]); ties by ascending index."""
    torch = _torch()
    lib = _ffi.load()
    dev = scores.device.index
    out_s = torch.empty(k, dtype=torch.float32, device=scores.device)
    out_i = torch.empty(k, dtype=torch.int64, device=scores.device)
    stream = torch.cuda.current_stream(scores.device).cuda_stream
    _ffi.check(
        lib.pmx_topk(
            scores.data_ptr(),
            indices.data_ptr() if indices is not None else None,
            scores.numel(),
            base_index,
            k,
            out_s.data_ptr(),
            out_i.data_ptr(),
            dev,
            ctypes.c_void_p(stream),
        )
    )
    return out_s, out_i


def screen(
    model,
    library,
    weights: dict[str, float] | None = None,
    topk: int | None = None,
    device=None,
    first: int = 0,
    count: int | None = None,
    index_base: int = 0,
    float64: bool = False,
) -> ScreeningResult:
    """Score ligands `[first, first + count)` of `library` against `model` on the GPU.

    `library` is a `DeviceLibrary` (already in HBM) or anything `as_packed_library` accepts.
    `index_base` is added to positions when reporting top-k indices (the shard's global offset).
    `float64`: scores as the float64 the reference returns (`pmx_score_f64`; `graph_match.py:109`) instead of its float32
    rounding; the device top-k ranks float32 values, so it is not offered together with `topk`."""
    torch = _torch()
    lib = _ffi.load()
    owned = None
    if not isinstance(library, DeviceLibrary):
        owned = library = DeviceLibrary(as_packed_library(library), device)
    dev = library.device
    mh = device_model(model, dev)
    if count is None:
        count = len(library) - first
    tdev = torch.device("cuda", dev)
    if float64 and topk is not None:
        raise ValueError("float64 scores are ranked by the caller (the device top-k ranks float32 values)")
    scores = torch.empty(count, dtype=torch.float64 if float64 else torch.float32, device=tdev)
    status = torch.empty(count, dtype=torch.int32, device=tdev)
    stream = torch.cuda.current_stream(tdev).cuda_stream
    try:
        _ffi.check(
            (lib.pmx_score_f64 if float64 else lib.pmx_score)(
                mh.handle, library.handle, _weights_array(weights), first, count,
                scores.data_ptr(), status.data_ptr(), ctypes.c_void_p(stream),
            )
        )
        result = ScreeningResult(scores=scores, status=status, first=first)
        if topk is not None:
            result.topk_scores, result.topk_indices = globals()["topk"](scores, int(topk), base_index=index_base + first)
    finally:
        if owned is not None:
            torch.cuda.synchronize(tdev)
            owned.close()
    return result


def score_one(model, ligand, weights: dict[str, float] | None = None, device=None) -> float:
    """`PharmacophoreModel._scoring` for one ligand: a Python float, like the reference returns."""
    packed = as_packed_library(ligand)
    if len(packed) != 1:
        raise ValueError("_scoring takes exactly one ligand")
    n, c, ncl = packed.header(0)
    if ncl == 0 and n == 0 and c > 0:
        return 0  # `GraphMatcher.run()` returns the int 0 for a ligand without clusters (graph_match.py:95-96)
    result = screen(model, packed, weights=weights, device=device, float64=True)  # (the reference returns a Python float: a float64)
    if int(result.status.cpu()[0]) != 0:
        raise ValueError(
            f"ligand outside the structural limits of the GPU engine (nodes={n}, conformers={c}); see include/pmx.h"
        )
    return float(result.scores.cpu()[0])


def last_score_stats() -> dict:
    st = _ffi.ScoreStats()
    _ffi.check(_ffi.load().pmx_score_stats_get(ctypes.byref(st)))
    return {name: (list(getattr(st, name)) if name == "dbg" else getattr(st, name)) for name, _ in st._fields_}


def release_workspaces(device=None) -> None:
    """Free the device buffers libpmx caches between scoring calls (`pmx_release_workspaces`)."""
    _ffi.check(_ffi.load().pmx_release_workspaces(_device_index(device)))


def set_profiling(enabled: bool) -> None:
    _ffi.check(_ffi.load().pmx_set_profiling(1 if enabled else 0))
