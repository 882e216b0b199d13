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

__all__ = ["Ligand", "perceive_features"]

_HALOGENS = (9, 17, 35, 53)


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


def _neighbors(ob, atom):
    return list(ob.OBAtomAtomIter(atom))


def _count(ob, atom, z):
    return sum(1 for n in _neighbors(ob, atom) if n.GetAtomicNum() == z)


def perceive_features(pbmol) -> tuple[list[int], list[list[int]], list[tuple]]:
    """Atomic numbers, heavy-atom neighbour lists and the typed feature list of a hydrogen-free pybel
    molecule: the rules of `ligand_utils.py:25-184`, emitted in the type order of `:80-88`."""
    pybel, ob = _openbabel()
    obmol = pbmol.OBMol
    atoms = list(ob.OBMolAtomIter(obmol))
    n = len(atoms)
    with_h = pbmol.clone
    with_h.OBMol.AddPolarHydrogens()  # donors are judged on the molecule with polar hydrogens (:30-34,46)
    atoms_h = list(ob.OBMolAtomIter(with_h.OBMol))[:n]

    z = [a.GetAtomicNum() for a in atoms]
    nbrs = [[m.GetIdx() - 1 for m in _neighbors(ob, a) if m.GetAtomicNum() != 1] for a in atoms]

    def nbr_idx(a, only=None):
        return tuple(m.GetIdx() - 1 for m in _neighbors(ob, a) if only is None or m.GetAtomicNum() == only)

    hydrophobic = [i for i, a in enumerate(atoms)
                   if z[i] == 6 and all(m.GetAtomicNum() in (1, 6) for m in _neighbors(ob, a))]            # :36-40
    acceptors = [i for i, a in enumerate(atoms) if z[i] not in _HALOGENS and a.IsHbondAcceptor()]           # :41-45
    donors = [i for i, a in enumerate(atoms_h) if a.IsHbondDonor()]                                           # :46
    rings = sorted(tuple(sorted(i - 1 for i in ring._path)) for ring in pbmol.sssr if ring.IsAromatic())      # :47-52

    cations: list[tuple] = []
    anions: list[tuple] = []
    for i, a in enumerate(atoms):  # single charged atoms first (:54-58)
        quaternary_n = z[i] == 7 and a.GetExplicitDegree() == 4 and _count(ob, a, 1) == 0                     # :94-103
        tertiary_n = z[i] == 7 and a.GetHyb() == 3 and a.GetHvyDegree() == 3                                  # :106-107
        sulfonium = z[i] == 16 and a.GetExplicitDegree() == 3 and _count(ob, a, 1) == 0                       # :110-118
        if quaternary_n or tertiary_n or sulfonium:
            cations.append((i, i))
    for i, a in enumerate(atoms):  # then charged groups (:61-76)
        ns = _neighbors(ob, a)
        guanidine = (z[i] == 6 and len(ns) > 0 and all(m.GetAtomicNum() == 7 for m in ns) and len(ns) == 3
                     and any(m.GetHvyDegree() == 1 for m in ns))                                              # :121-133
        phosphate = z[i] == 15 and all(m.GetAtomicNum() == 8 for m in ns)                                     # :156-162
        sulfate = z[i] == 16 and _count(ob, a, 8) == 4                                                        # :146-153
        sulfonic = z[i] == 16 and _count(ob, a, 8) == 3                                                       # :136-143
        carboxylate = z[i] == 6 and _count(ob, a, 8) == 2 and _count(ob, a, 6) == 1                           # :165-175
        if guanidine:
            cations.append(((i,) + nbr_idx(a, 7), i))
        elif phosphate or sulfate:
            anions.append(((i,) + nbr_idx(a), i))
        elif sulfonic:
            anions.append(((i,) + nbr_idx(a, 8), i))
        elif carboxylate:
            oxygens = nbr_idx(a, 8)
            anions.append(((i,) + oxygens, oxygens))
    halogens = [i for i, a in enumerate(atoms) if z[i] in _HALOGENS and _count(ob, a, 6) > 0]                 # :78,178-184

    features: list[tuple] = []
    features += [("Hydrophobic", i, i) for i in hydrophobic]
    features += [("Aromatic", r, r) for r in rings]
    features += [("Cation", at, ce) for at, ce in cations]
    features += [("Anion", at, ce) for at, ce in anions]
    features += [("HBond_donor", i, i) for i in donors]
    features += [("HBond_acceptor", i, i) for i in acceptors]
    features += [("Halogen", i, i) for i in halogens]
    return z, nbrs, features


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
        z, nbrs, feats = perceive_features(self.pbmol)
        self.features = LigandFeatures(z, nbrs, feats, pos)

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
