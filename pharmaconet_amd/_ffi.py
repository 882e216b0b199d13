"""ctypes binding of libpmx.so (include/pmx.h). The product path fails loudly when the HIP
library is missing or cannot be loaded: there is no CPU fallback in this package."""

from __future__ import annotations

import ctypes
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libpmx.so"

NUM_TYPES = 7
COMM_ID_BYTES = 128


class PmxError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [
        ("n_nodes", ctypes.c_int32),
        ("n_clusters", ctypes.c_int32),
        ("node_type", ctypes.c_void_p),
        ("edge_mean", ctypes.c_void_p),
        ("edge_std", ctypes.c_void_p),
        ("cluster_nodes", ctypes.c_void_p),
        ("cluster_typemask", ctypes.c_void_p),
        ("cluster_center", ctypes.c_void_p),
        ("cluster_size", ctypes.c_void_p),
    ]


class LibraryView(ctypes.Structure):
    _fields_ = [
        ("n_ligands", ctypes.c_uint64),
        ("offsets", ctypes.c_void_p),
        ("data", ctypes.c_void_p),
        ("on_device", ctypes.c_int32),
    ]


class LibraryInfo(ctypes.Structure):
    _fields_ = [
        ("n_ligands", ctypes.c_uint64),
        ("n_bytes", ctypes.c_uint64),
        ("total_conformers", ctypes.c_uint64),
        ("max_nodes", ctypes.c_int32),
        ("max_conformers", ctypes.c_int32),
        ("max_clusters", ctypes.c_int32),
        ("n_unsupported", ctypes.c_int32),
    ]


class FeatureBatch(ctypes.Structure):
    _fields_ = [
        ("n_mols", ctypes.c_uint64),
        ("atom_off", ctypes.c_void_p),
        ("atomic_num", ctypes.c_void_p),
        ("nbr_off", ctypes.c_void_p),
        ("nbr", ctypes.c_void_p),
        ("feat_off", ctypes.c_void_p),
        ("feat_type", ctypes.c_void_p),
        ("feat_flags", ctypes.c_void_p),
        ("feat_atom_off", ctypes.c_void_p),
        ("feat_atoms", ctypes.c_void_p),
        ("feat_center_off", ctypes.c_void_p),
        ("feat_centers", ctypes.c_void_p),
        ("n_conf", ctypes.c_void_p),
        ("pos_off", ctypes.c_void_p),
        ("positions", ctypes.c_void_p),
    ]


class AtomBatch(ctypes.Structure):
    _fields_ = [("n_mols", ctypes.c_uint64)] + [(n, ctypes.c_void_p) for n in (
        "atom_off", "atomic_num", "explicit_degree", "heavy_degree", "hyb", "h_count", "flags", "nbr_off", "nbr", "ring_off", "ring_atom_off", "ring_atoms")]


class ScoreStats(ctypes.Structure):
    _fields_ = (
        [("ms_total", ctypes.c_double), ("ms_ligand", ctypes.c_double), ("ms_tasks", ctypes.c_double)]
        + [(n, ctypes.c_uint64) for n in (
            "ligands_last", "n_frames", "n_passes", "n_items", "n_exact_cells", "n_heavy", "n_tasks", "n_exported",
            "n_slice_overflow", "n_probes", "n_probe_passes", "max_passes", "queue_overflow", "arena_bytes",
            "ticks_scan", "ticks_tables", "ticks_bounds", "ticks_walk", "ticks_alive", "n_exact_values")]
        + [("dbg", ctypes.c_uint64 * 8), ("n_path_bounds", ctypes.c_uint64), ("n_path_drops", ctypes.c_uint64),
           ("n_dead_entries", ctypes.c_uint64), ("arena_capacity", ctypes.c_uint64)]
    )


# name -> (restype, argtypes); every symbol include/pmx.h declares
SIGNATURES = {
    "pmx_last_error": (ctypes.c_char_p, []),
    "pmx_version": (ctypes.c_int, []),
    "pmx_model_create": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_model_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_library_upload": (ctypes.c_int, [ctypes.POINTER(LibraryView), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_library_info_get": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(LibraryInfo)]),
    "pmx_library_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_score": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_uint64, ctypes.c_uint64,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_score_multi": (
        ctypes.c_int,
        [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
         ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_score_f64": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_uint64, ctypes.c_uint64,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_score_multi_f64": (
        ctypes.c_int,
        [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
         ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_topk": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "pmx_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "pmx_comm_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_comm_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "pmx_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_topk_allgather": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_release_workspaces": (ctypes.c_int, [ctypes.c_int]),
    "pmx_density_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pmx_density_labels": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "pmx_density_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pmx_density_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pmx_pack_features": (
        ctypes.c_int,
        [ctypes.POINTER(FeatureBatch), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p],
    ),
    "pmx_pack_features_device": (
        ctypes.c_int,
        [ctypes.POINTER(FeatureBatch), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
         ctypes.c_void_p],
    ),
    "pmx_perceive_features": (
        ctypes.c_int,
        [ctypes.POINTER(AtomBatch), ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_uint64] * 3 + [ctypes.POINTER(ctypes.c_uint64)] * 3 + [ctypes.c_void_p],
    ),
    "pmx_sdf_heavy_atoms": (
        ctypes.c_int,
        [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
         ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_mol2_heavy_atoms": (
        ctypes.c_int,
        [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
         ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "pmx_score_stats_get": (ctypes.c_int, [ctypes.POINTER(ScoreStats)]),
    "pmx_set_profiling": (ctypes.c_int, [ctypes.c_int]),
}

_lib = None


def load(need_torch: bool = True) -> ctypes.CDLL:
    """Load libpmx.so from the package directory (built by `python -m pharmaconet_amd.build`)."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's); import it first so that the
        # process has ONE HIP runtime - the one that owns torch's device buffers and streams.
        # (Host-only entry points - the packer - are used by worker processes that never touch a GPU.)
        if need_torch:
            import torch  # noqa: F401

        import os

        path = Path(os.environ["PMX_LIBPMX"]) if os.environ.get("PMX_LIBPMX") else LIB_PATH  # (A/B builds: tools/build_variant.py)
        if not path.exists():
            raise PmxError(
                f"{path} is missing: build the HIP extension with `python -m pharmaconet_amd.build` "
                "(hipcc, gfx950). There is no CPU scoring path."
            )
        try:
            lib = ctypes.CDLL(str(path))
        except OSError as e:  # e.g. libamdhip64 not found
            raise PmxError(f"cannot load {path}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


_pack_lib: ctypes.CDLL | None = None


def load_packer() -> ctypes.CDLL:
    """libpmx_pack.so: `pmx_pack_features` and the SD-file coordinate reader alone, for host processes that read and pack
    libraries (no torch, no HIP runtime)."""
    global _pack_lib
    if _pack_lib is None:
        path = LIB_PATH.with_name("libpmx_pack.so")
        if not path.exists():
            raise PmxError(f"{path} is missing: build with `python -m pharmaconet_amd.build`")
        lib = ctypes.CDLL(str(path))
        for name in ("pmx_pack_features", "pmx_perceive_features", "pmx_sdf_heavy_atoms", "pmx_mol2_heavy_atoms", "pmx_last_error", "pmx_version"):
            restype, argtypes = SIGNATURES[name]
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _pack_lib = lib
    return _pack_lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().pmx_last_error()
        raise PmxError(f"libpmx error {rc}: {msg.decode() if msg else '?'}")
