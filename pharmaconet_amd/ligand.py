"""`Ligand`: one molecule with its conformers, turned into the packer's input (`LigandFeatures`).

Host-side mirror of `src/pmnet/scoring/ligand.py:16-107` (molecule + `[N_atoms, N_conformers, 3]` coordinates)
and of the perception in `src/pmnet/scoring/ligand_utils.py:25-184`. Perception needs OpenBabel exactly as
in the reference (`openbabel-wheel`, `pyproject.toml:34`); it is an optional dependency here and this
module raises ImportError with instructions when it is absent. Everything after perception - the graph
builder (`library.cluster_ligand`), packing and scoring - has no OpenBabel dependency.

Parity: OpenBabel is not installed in the build image, so real molecule files cannot be read here. The rule logic of
this file IS pinned: `tests/golden/make_golden_perception.py` runs the reference's own `get_pharmacophore_nodes`,
`Ligand.__init__`, `Ligand.load_from_file` and `scoring_pbmol` / `scoring_file` on 600 described molecules through a
stand-in for the OpenBabel calls (`tests/fake_openbabel.py`), and `tests/test_perception.py` holds this file to those
outputs (feature lists, packed records, scores). OpenBabel's own atom typing (`IsHbondAcceptor`, hybridisation, ring
perception) is its business in both code bases and is not pinned.
"""

from __future__ import annotations

import itertools
import os
from pathlib import Path

import numpy as np

from .library import LigandFeatures

__all__ = ["Ligand", "perceive_features", "toolkit_answers", "perceive_batch", "features_of"]

def _openbabel():
    try:
        from openbabel import pybel  # type: ignore
        from openbabel.pybel import ob  # type: ignore
    except Exception as e:  # pragma: no cover - depends on the environment
        raise ImportError(
            "reading molecule files / SMILES needs OpenBabel (pip install openbabel-wheel), as in the reference; "
            "alternatively score pre-packed libraries (pharmaconet_amd.library) which need no chemistry toolkit"
        ) from e
    return pybel, ob


def _ffi_error():
    from ._ffi import PmxError

    return PmxError


def toolkit_answers(pbmol) -> dict[str, np.ndarray]:
    """What only the chemistry toolkit can say about a hydrogen-free pybel molecule, as the flat per-atom arrays of
    `pmx_atom_batch` (include/pmx.h): element, degrees, hybridisation, acceptor / donor flags, heavy-atom neighbours in
    `OBAtomAtomIter` order, the aromatic rings of the SSSR. One pass over the molecule's atoms; every rule of
    `ligand_utils.py:25-184` is then decided in native code (`perceive_batch`)."""
    pybel, ob = _openbabel()
    obmol = pbmol.OBMol
    atoms = list(ob.OBMolAtomIter(obmol))
    n = len(atoms)
    with_h = pbmol.clone
    with_h.OBMol.AddPolarHydrogens()  # donors are judged on the molecule with polar hydrogens (:30-34,46)
    atoms_h = list(ob.OBMolAtomIter(with_h.OBMol))[:n]
    z = np.fromiter((a.GetAtomicNum() for a in atoms), dtype=np.uint8, count=n)
    nbr_off = np.zeros(n + 1, dtype=np.uint64)
    nbr: list[int] = []
    h_count = np.zeros(n, dtype=np.uint8)
    for i, a in enumerate(atoms):
        for m in ob.OBAtomAtomIter(a):
            if m.GetAtomicNum() == 1:
                h_count[i] += 1
            else:
                nbr.append(m.GetIdx() - 1)
        nbr_off[i + 1] = len(nbr)
    rings = [[i - 1 for i in ring._path] for ring in pbmol.sssr if ring.IsAromatic()]
    ring_atom_off = np.zeros(len(rings) + 1, dtype=np.uint64)
    if rings:
        ring_atom_off[1:] = np.cumsum([len(r) for r in rings])
    return dict(
        atomic_num=z,
        explicit_degree=np.fromiter((a.GetExplicitDegree() for a in atoms), dtype=np.uint8, count=n),
        heavy_degree=np.fromiter((a.GetHvyDegree() for a in atoms), dtype=np.uint8, count=n),
        hyb=np.fromiter((a.GetHyb() for a in atoms), dtype=np.uint8, count=n),
        h_count=h_count,
        flags=np.fromiter(((1 if a.IsHbondAcceptor() else 0) | (2 if ah.IsHbondDonor() else 0) for a, ah in zip(atoms, atoms_h)), dtype=np.uint8, count=n),
        nbr_off=nbr_off, nbr=np.asarray(nbr, dtype=np.int32),
        ring_atom_off=ring_atom_off, ring_atoms=np.asarray([i for r in rings for i in r], dtype=np.int32),
    )


def perceive_batch(answers: list[dict[str, np.ndarray]], positions: list[np.ndarray] | None = None, threads: int = 1) -> dict[str, np.ndarray]:
    """`get_pharmacophore_nodes` (`ligand_utils.py:25-184`) for a batch of molecules in native code (`pmx_perceive_features`,
    csrc/pmx_perceive.cpp): the toolkit's answers of `toolkit_answers` in, the flat feature batch out - with `positions`
    (float32 `[n_atoms, n_conformers, 3]` per molecule) the complete input of `library.pack_features_native`, so a library goes
    from toolkit answers to packed records without a Python loop over molecules, atoms or features."""
    import ctypes

    from . import _ffi

    lib = _ffi.load_packer()
    n = len(answers)

    def cat(key, dtype):
        return np.ascontiguousarray(np.concatenate([a[key] for a in answers]) if n else np.zeros(0, dtype), dtype=dtype)

    n_atoms = np.array([len(a["atomic_num"]) for a in answers], dtype=np.uint64)
    atom_off = np.zeros(n + 1, dtype=np.uint64)
    atom_off[1:] = np.cumsum(n_atoms)
    nbr_len = np.array([len(a["nbr"]) for a in answers], dtype=np.uint64)
    nbr_base = np.zeros(n + 1, dtype=np.uint64)
    nbr_base[1:] = np.cumsum(nbr_len)
    nbr_off = np.zeros(int(atom_off[-1]) + 1, dtype=np.uint64)
    for i, a in enumerate(answers):  # (per molecule, vectorised inside: offsets become absolute)
        nbr_off[int(atom_off[i]) : int(atom_off[i + 1]) + 1] = a["nbr_off"] + nbr_base[i]
    n_rings = np.array([len(a["ring_atom_off"]) - 1 for a in answers], dtype=np.uint64)
    ring_off = np.zeros(n + 1, dtype=np.uint64)
    ring_off[1:] = np.cumsum(n_rings)
    ring_atom_off = np.zeros(int(ring_off[-1]) + 1, dtype=np.uint64)
    base = 0
    for i, a in enumerate(answers):
        r0, r1 = int(ring_off[i]), int(ring_off[i + 1])
        ring_atom_off[r0 : r1 + 1] = a["ring_atom_off"] + np.uint64(base)
        base += int(a["ring_atom_off"][-1])
    arrays = dict(
        atom_off=atom_off, atomic_num=cat("atomic_num", np.uint8), explicit_degree=cat("explicit_degree", np.uint8),
        heavy_degree=cat("heavy_degree", np.uint8), hyb=cat("hyb", np.uint8), h_count=cat("h_count", np.uint8), flags=cat("flags", np.uint8),
        nbr_off=nbr_off, nbr=cat("nbr", np.int32), ring_off=ring_off, ring_atom_off=ring_atom_off, ring_atoms=cat("ring_atoms", np.int32),
    )
    batch = _ffi.AtomBatch(n, *(arrays[k].ctypes.data for k in (
        "atom_off", "atomic_num", "explicit_degree", "heavy_degree", "hyb", "h_count", "flags", "nbr_off", "nbr", "ring_off", "ring_atom_off", "ring_atoms")))
    feat_off = np.zeros(n + 1, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    counts = [ctypes.c_uint64(0) for _ in range(3)]

    def call(*bufs, caps=(0, 0, 0)):
        rc = lib.pmx_perceive_features(ctypes.byref(batch), int(threads), feat_off.ctypes.data, *bufs, *caps, *(ctypes.byref(c) for c in counts), status.ctypes.data)
        if rc != 0:
            raise _ffi.PmxError(f"pmx_perceive_features failed ({rc}): {lib.pmx_last_error().decode()}")

    call(*([None] * 6))  # counts
    nf, na, nc = (int(c.value) for c in counts)
    feat_type, feat_flags = np.zeros(nf, np.uint8), np.zeros(nf, np.uint8)
    fa_off, fc_off = np.zeros(nf + 1, np.uint64), np.zeros(nf + 1, np.uint64)
    fa, fc = np.zeros(max(na, 1), np.int32), np.zeros(max(nc, 1), np.int32)
    call(feat_type.ctypes.data, feat_flags.ctypes.data, fa_off.ctypes.data, fa.ctypes.data, fc_off.ctypes.data, fc.ctypes.data, caps=(nf, na, nc))
    if status.any():
        raise ValueError(f"perceive_batch: malformed toolkit answers for molecule(s) {np.flatnonzero(status).tolist()[:8]}")
    out = dict(
        atom_off=atom_off, atomic_num=arrays["atomic_num"], nbr_off=nbr_off, nbr=arrays["nbr"], feat_off=feat_off, feat_type=feat_type,
        feat_flags=feat_flags, feat_atom_off=fa_off, feat_atoms=fa[:na], feat_center_off=fc_off, feat_centers=fc[:nc],
    )
    if positions is not None:
        flat_pos = [np.ascontiguousarray(p, dtype=np.float32) for p in positions]
        pos_off = np.zeros(n + 1, dtype=np.uint64)
        pos_off[1:] = np.cumsum([p.size for p in flat_pos])
        out.update(n_conf=np.array([int(p.shape[1]) if p.ndim == 3 else 0 for p in flat_pos], dtype=np.int32), pos_off=pos_off,
                   positions=np.concatenate([p.reshape(-1) for p in flat_pos]) if flat_pos else np.zeros(0, np.float32))
    return out


def features_of(flat: dict[str, np.ndarray], i: int) -> list[tuple]:
    """Molecule `i` of a flat feature batch as the reference's `pharmacophore_list` entries `(type, atom_indices, center_indices)`
    - ints or tuples as the reference holds them (an int and a 1-tuple are different node keys, ligand.py:137)."""
    from .constants import TYPE_NAMES

    out = []
    for k in range(int(flat["feat_off"][i]), int(flat["feat_off"][i + 1])):
        atoms = flat["feat_atoms"][int(flat["feat_atom_off"][k]) : int(flat["feat_atom_off"][k + 1])].tolist()
        centers = flat["feat_centers"][int(flat["feat_center_off"][k]) : int(flat["feat_center_off"][k + 1])].tolist()
        fl = int(flat["feat_flags"][k])
        out.append((TYPE_NAMES[int(flat["feat_type"][k])], tuple(atoms) if fl & 1 else atoms[0], tuple(centers) if fl & 2 else centers[0]))
    return out


def perceive_features(pbmol) -> tuple[list[int], list[list[int]], list[tuple]]:
    """Atomic numbers, heavy-atom neighbour lists and the typed feature list of a hydrogen-free pybel molecule: the toolkit's
    answers (`toolkit_answers`) through the native rules of `ligand_utils.py:25-184` (`perceive_batch`), emitted in the type
    order of `:80-88`."""
    ans = toolkit_answers(pbmol)
    flat = perceive_batch([ans])
    z = [int(x) for x in ans["atomic_num"]]
    off = ans["nbr_off"].astype(np.int64)
    nbrs = [ans["nbr"][off[i] : off[i + 1]].tolist() for i in range(len(z))]
    return z, nbrs, features_of(flat, 0)


class Ligand:
    """A molecule and its conformers (`ligand.py:16-61`); `.features` is what gets packed and scored."""

    def __init__(self, pbmol, atom_positions, conformer_axis: int | None = None, _unsafe: bool = False):
        _openbabel()
        self.pbmol = pbmol if _unsafe else pbmol.clone
        self.pbmol.removeh()
        self.num_atoms = self.pbmol.OBMol.NumAtoms()
        self.num_rotatable_bonds = pbmol.OBMol.NumRotors()
        if isinstance(atom_positions, list):  # list of [N_atoms, 3] per conformer (:46-47)
            pos = np.stack(atom_positions, axis=1).astype(np.float32)
        else:
            pos = np.asarray(atom_positions, dtype=np.float32)
            if conformer_axis in (0, None):  # (N_conformers, N_atoms, 3) -> atoms first (:50-51)
                pos = np.ascontiguousarray(np.moveaxis(pos, 0, 1))
        assert self.num_atoms == pos.shape[0]
        self.atom_positions = pos
        self.num_conformers = int(pos.shape[1])
        self.answers = toolkit_answers(self.pbmol)  # everything that is asked of the toolkit; the rules run in native code
        self._features = None

    @property
    def features(self) -> LigandFeatures:
        """The packer's input for this one molecule (`perceive_batch` takes many molecules' `answers` at once)."""
        if self._features is None:
            ans = self.answers
            off = ans["nbr_off"].astype(np.int64)
            nbrs = [ans["nbr"][off[i] : off[i + 1]].tolist() for i in range(len(ans["atomic_num"]))]
            feats = features_of(perceive_batch([ans]), 0)
            self._features = LigandFeatures([int(x) for x in ans["atomic_num"]], nbrs, feats, self.atom_positions)
        return self._features

    @classmethod
    def load_from_file(cls, filename: str | Path, num_conformers: int | None = None) -> "Ligand":
        """Every record of the file is one conformer of the same molecule (`ligand.py:63-84`)."""
        pybel, _ = _openbabel()
        assert filename is not None
        extension = os.path.splitext(filename)[1]
        assert extension in [".sdf", ".pdb", ".mol2"]
        reader = pybel.readfile(extension[1:], str(filename))
        if num_conformers is not None:
            assert num_conformers > 0
            reader = itertools.islice(reader, num_conformers)
        if extension in (".sdf", ".mol2"):
            # Features come from the first record alone; of the others only heavy-atom coordinates are used. Those are read by the
            # native SD / mol2 reader (csrc/pmx_sdf.cpp) instead of one toolkit molecule + a Python loop over its atoms per record.
            # Anything it does not accept - or records that disagree with the toolkit's view of the first one - goes the
            # reference's way below.
            fast = cls._from_sdf_fast(reader, filename, num_conformers)
            if fast is not None:
                return fast
            reader = pybel.readfile(extension[1:], str(filename))
            if num_conformers is not None:
                reader = itertools.islice(reader, num_conformers)
        mols = list(reader)
        base = mols[0]
        base.removeh()
        num_atoms = len(base.atoms)
        positions = []
        for mol in mols:
            mol.removeh()
            assert len(mol.atoms) == num_atoms
            positions.append([atom.coords for atom in mol.atoms])
        return cls(base, [np.asarray(p, dtype=np.float32) for p in positions], _unsafe=True)

    @classmethod
    def _from_sdf_fast(cls, reader, filename, num_conformers):
        from .sdf import SdfError, conformer_positions

        try:
            z, pos = conformer_positions(filename, num_conformers)
        except (SdfError, OSError, _ffi_error()):
            return None
        base = next(iter(reader), None)
        if base is None:
            return None
        base.removeh()
        if len(base.atoms) != len(z) or [a.GetAtomicNum() for a in _openbabel()[1].OBMolAtomIter(base.OBMol)] != [int(x) for x in z]:
            return None
        return cls(base, pos, conformer_axis=1, _unsafe=True)

    @classmethod
    def load_from_smiles(cls, smiles: str, num_conformers: int) -> "Ligand":
        """Embed conformers with RDKit's srETKDGv3 and read them back (`ligand.py:86-107`)."""
        import tempfile

        try:
            from rdkit import Chem  # type: ignore
            from rdkit.Chem import rdDistGeom  # type: ignore
        except Exception as e:  # pragma: no cover
            raise ImportError("scoring_smiles needs RDKit for conformer embedding, as in the reference") from e
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "conformers.sdf")
            mol = Chem.AddHs(Chem.MolFromSmiles(smiles))
            rdDistGeom.EmbedMultipleConfs(mol, num_conformers, params=rdDistGeom.srETKDGv3())
            with Chem.SDWriter(path) as w:
                for i in range(mol.GetNumConformers()):
                    w.write(mol, confId=i)
            return cls.load_from_file(path)
