"""Conformer coordinates of an SD or Tripos mol2 file through the native readers (`pmx_sdf_heavy_atoms`,
`pmx_mol2_heavy_atoms`, csrc/pmx_sdf.cpp).

`Ligand.load_from_file` (reference `src/pmnet/scoring/ligand.py:63-84`) perceives features on the first record of a
multi-conformer file and takes nothing but heavy-atom coordinates from the others - through one OpenBabel molecule
object per record and a Python loop over its atoms. This module reads those coordinates natively: element and float32
position of every non-hydrogen atom, record by record, in file order (what `pbmol.removeh()` leaves). No chemistry
toolkit is involved; the host-only `libpmx_pack.so` carries the reader, so no GPU runtime is loaded either.
"""

from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

from . import _ffi

__all__ = ["SdfError", "read_heavy_atoms", "conformer_positions"]


class SdfError(ValueError):
    """The file is not an SD file this reader understands (the record's index is in the message)."""


def read_heavy_atoms(source: str | Path | bytes, max_records: int | None = None, fmt: str | None = None) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(heavy atoms per record int32[R], atomic numbers uint8[sum], positions float32[sum, 3]) of an SD or mol2 file (path or
    bytes). `fmt`: "sdf" or "mol2"; by default the path's extension decides (bytes: "sdf")."""
    if fmt is None:
        fmt = "mol2" if not isinstance(source, (bytes, bytearray)) and str(source).lower().endswith(".mol2") else "sdf"
    if fmt not in ("sdf", "mol2"):
        raise ValueError(f"no native reader for format {fmt!r}")
    text = source if isinstance(source, (bytes, bytearray)) else Path(source).read_bytes()
    text = bytes(text)
    lib = _ffi.load_packer()
    reader = lib.pmx_sdf_heavy_atoms if fmt == "sdf" else lib.pmx_mol2_heavy_atoms
    n_rec, n_atoms = ctypes.c_uint64(0), ctypes.c_uint64(0)
    limit = int(max_records or 0)
    rc = reader(text, len(text), limit, 0, 0, ctypes.byref(n_rec), ctypes.byref(n_atoms), None, None, None)
    if rc != 0:
        raise SdfError(f"record {n_rec.value} is not a {fmt} record this reader understands")
    per = np.zeros(n_rec.value, dtype=np.int32)
    z = np.zeros(n_atoms.value, dtype=np.uint8)
    xyz = np.zeros((n_atoms.value, 3), dtype=np.float32)
    rc = reader(text, len(text), limit, per.size, z.size, ctypes.byref(n_rec), ctypes.byref(n_atoms),
                per.ctypes.data, z.ctypes.data, xyz.ctypes.data)
    if rc != 0:
        raise SdfError(f"record {n_rec.value} is not a {fmt} record this reader understands")
    return per, z, xyz


def conformer_positions(source: str | Path | bytes, max_records: int | None = None, fmt: str | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Every record as a conformer of one molecule: (atomic numbers uint8[N], positions float32[N, C, 3]).
    Raises `SdfError` unless all records have the same heavy atoms in the same order (`ligand.py:80-83` asserts the count)."""
    per, z, xyz = read_heavy_atoms(source, max_records, fmt)
    if per.size == 0:
        raise SdfError("no record in the file")
    n = int(per[0])
    if np.any(per != n):
        raise SdfError("the records of the file do not have the same number of heavy atoms")
    z = z.reshape(per.size, n)
    if np.any(z != z[0]):
        raise SdfError("the records of the file do not list the same elements in the same order")
    return z[0].copy(), np.ascontiguousarray(np.moveaxis(xyz.reshape(per.size, n, 3), 0, 1))
