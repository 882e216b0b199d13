"""ctypes binding of oracle/libpmx_oracle.so (the C restatement of `GraphMatcher.run()`).

TEST INFRASTRUCTURE, NOT PRODUCT: imported by tests/, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of bench.py only. Parity status: pinned against reference outputs
(tests/golden/, see pmx_oracle.c).
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libpmx_oracle.so"


class OracleModel(ctypes.Structure):
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


RESULT_DTYPE = np.dtype(
    [
        ("score", "<f8"),
        ("n_levels", "<i4"),
        ("_pad", "<i4"),
        ("n_tree", "<i8"),
        ("n_leaf", "<i8"),
        ("s_sum", "<f8"),
        ("p_sum", "<f8"),
        ("p_invalid", "<i8"),
        ("p_entries", "<i8"),
        ("n_terms", "<i8"),
    ]
)


def build(force: bool = False) -> Path:
    import fcntl

    src = HERE / "pmx_oracle.c"

    def stale():
        return force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime

    if stale():
        with open(HERE / ".build.lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.run(["make", "-C", str(HERE), "-B", "libpmx_oracle.so"], check=True, capture_output=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(LIB_PATH))
        _lib.oracle_score.restype = ctypes.c_int
        _lib.oracle_score.argtypes = [
            ctypes.POINTER(OracleModel),
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_uint64,
            ctypes.c_uint64,
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_int,
        ]
        _lib.oracle_score_variant.restype = ctypes.c_int
        _lib.oracle_score_variant.argtypes = list(_lib.oracle_score.argtypes) + [ctypes.c_int]
    return _lib


def oracle_score(flat_model, library, weights7, first: int = 0, count: int | None = None, num_threads: int = 1,
                 with_stats: bool = False, variant: str = "numpy"):
    """Score `count` ligands of a `PackedLibrary` against a `FlatModel`; returns float64 scores
    (and the per-ligand statistics record array when `with_stats`). `variant`: "numpy" (the reference's
    match_utils.py, what the golden vectors pin) or "numba" (restatement of match_utils_numba.py, unpinned)."""
    lib = _load()
    if count is None:
        count = len(library) - first
    keep = dict(
        node_type=np.ascontiguousarray(flat_model.node_type, dtype=np.uint8),
        edge_mean=np.ascontiguousarray(flat_model.edge_mean, dtype=np.float32),
        edge_std=np.ascontiguousarray(flat_model.edge_std, dtype=np.float32),
        cluster_nodes=np.ascontiguousarray(flat_model.cluster_nodes, dtype=np.uint64),
        cluster_typemask=np.ascontiguousarray(flat_model.cluster_typemask, dtype=np.uint8),
        cluster_center=np.ascontiguousarray(flat_model.cluster_center, dtype=np.float64),
        cluster_size=np.ascontiguousarray(flat_model.cluster_size, dtype=np.float64),
    )
    model = OracleModel(
        flat_model.num_nodes,
        flat_model.num_clusters,
        *(keep[name].ctypes.data for name in (
            "node_type", "edge_mean", "edge_std", "cluster_nodes", "cluster_typemask", "cluster_center", "cluster_size")),
    )
    offsets = np.ascontiguousarray(library.offsets, dtype=np.uint64)
    data = np.ascontiguousarray(library.data, dtype=np.uint8)
    w = np.ascontiguousarray(weights7, dtype=np.float32)
    assert w.shape == (7,)
    scores = np.zeros(count, dtype=np.float64)
    stats = np.zeros(count, dtype=RESULT_DTYPE) if with_stats else None
    rc = lib.oracle_score_variant(
        ctypes.byref(model),
        offsets.ctypes.data,
        data.ctypes.data,
        first,
        count,
        w.ctypes.data,
        scores.ctypes.data,
        stats.ctypes.data if stats is not None else None,
        int(num_threads),
        {"numpy": 0, "numba": 1}[variant],
    )
    if rc != 0:
        raise RuntimeError(f"oracle_score failed ({rc})")
    return (scores, stats) if with_stats else scores
