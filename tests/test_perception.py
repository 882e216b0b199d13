"""Ligand perception (`pharmaconet_amd/ligand.py`) against the reference's own (`ligand_utils.py:25-184`,
`ligand.py:16-84`) on 600 described molecules: tests/golden/make_golden_perception.py ran the reference on them with
tests/fake_openbabel.py standing in for OpenBabel (absent from the image); here the same stand-in feeds this
repository's perception. Pins the rule logic - which atoms become which pharmacophore features, in which order, with
which atom / centre indices - not OpenBabel's own atom typing."""

import gzip
import json

import numpy as np
import pytest

import fake_openbabel
from conftest import GOLDEN


@pytest.fixture(scope="module", autouse=True)
def _openbabel_stand_in():
    """`import openbabel` resolves to the stand-in while this module's tests run, and to nothing again afterwards."""
    import sys

    fake_openbabel.install()
    yield
    for name in ("openbabel", "openbabel.pybel", "openbabel.pybel.ob"):
        if getattr(sys.modules.get(name), "__fake__", False) or name != "openbabel":
            sys.modules.pop(name, None)


@pytest.fixture(scope="module")
def golden():
    with gzip.open(GOLDEN / "perception.json.gz", "rt") as f:
        return json.load(f)


def _plain(x):
    return int(x) if isinstance(x, (int, np.integer)) else [int(i) for i in x]


def test_every_rule_branch_is_in_the_fixture(golden):
    assert len(golden["molecules"]) >= 500
    assert all(v > 0 for v in golden["branches"].values()), golden["branches"]
    assert set(golden["features_by_type"]) == {"Hydrophobic", "Aromatic", "Cation", "Anion", "HBond_donor", "HBond_acceptor", "Halogen"}
    assert all(v > 100 for v in golden["features_by_type"].values())


def test_perceive_features_matches_the_reference(golden):
    from pharmaconet_amd.ligand import perceive_features

    for i, (desc, want) in enumerate(zip(golden["molecules"], golden["reference"])):
        pb = fake_openbabel.Molecule(desc)
        pb.removeh()
        z, nbrs, feats = perceive_features(pb)
        got = [[t, _plain(a), _plain(c)] for t, a, c in feats]
        assert got == want, f"molecule {i}"
        assert z == desc["z"]
        adj = [[] for _ in z]
        for a, b in desc["bonds"]:
            adj[a].append(b)
            adj[b].append(a)
        assert nbrs == adj  # neighbour order = the order OBAtomAtomIter yields them


def test_ligand_packs_to_the_reference_graph(golden):
    """`Ligand(pbmol, positions)` -> features -> packed record == the record extracted from the reference's LigandGraph
    built by the reference's `Ligand.__init__` on the same molecule (perception + graph builder together)."""
    from pharmaconet_amd.library import pack_ligand
    from pharmaconet_amd.ligand import Ligand

    e2e = np.load(GOLDEN / "perception_e2e.npz")
    off = np.concatenate([[0], np.cumsum(e2e["record_len"])])
    for i in range(len(e2e["record_len"])):
        desc = golden["molecules"][i]
        lig = Ligand(fake_openbabel.Molecule(desc), np.asarray(desc["coords"], dtype=np.float32), conformer_axis=0)
        assert lig.num_conformers == len(desc["coords"]) and lig.num_atoms == len(desc["z"])
        assert lig.num_rotatable_bonds == desc["rotors"]
        rec = pack_ligand(lig.features)
        assert bytes(rec) == e2e["records"][off[i]:off[i + 1]].tobytes(), f"molecule {i}"


def test_load_from_file_takes_records_as_conformers(golden, tmp_path):
    from pharmaconet_amd.library import pack_ligand
    from pharmaconet_amd.ligand import Ligand

    for i in (0, 3, 5):
        desc = golden["molecules"][i]
        path = tmp_path / f"mol{i}.sdf"
        path.write_text(json.dumps(desc))
        a = Ligand.load_from_file(path)
        b = Ligand(fake_openbabel.Molecule(desc), np.asarray(desc["coords"], dtype=np.float32), conformer_axis=0)
        assert bytes(pack_ligand(a.features)) == bytes(pack_ligand(b.features))
        if len(desc["coords"]) > 1:
            one = Ligand.load_from_file(path, num_conformers=1)
            assert one.num_conformers == 1
    with pytest.raises(AssertionError):
        Ligand.load_from_file(tmp_path / "mol.xyz")


@pytest.mark.gpu
def test_scoring_pbmol_and_scoring_file_end_to_end(golden, tmp_path):
    """`PharmacophoreModel.scoring_pbmol` / `scoring_file` (pharmacophore_model.py:60-99) on the described molecules: perception,
    graph builder, packing, GPU scoring - against the floats the reference's own scoring_pbmol / scoring_file returned."""
    from pharmaconet_amd import PharmacophoreModel

    model = PharmacophoreModel.load(GOLDEN / "model_6oim_like.pm")
    e2e = np.load(GOLDEN / "perception_e2e.npz")
    worst = 0.0
    for i, want in enumerate(e2e["score"]):
        desc = golden["molecules"][i]
        got = model.scoring_pbmol(fake_openbabel.Molecule(desc), np.asarray(desc["coords"], dtype=np.float32), conformer_axis=0)
        assert isinstance(got, float)
        err = abs(got - want) / max(abs(want), 1e-30) if want else abs(got)
        worst = max(worst, err)
        assert err < 2e-6, f"molecule {i}: {got} vs {want}"
    for i, want in enumerate(e2e["file_score"]):
        path = tmp_path / f"m{i}.sdf"
        path.write_text(json.dumps(golden["molecules"][i]))
        got = model.scoring_file(path)
        assert abs(got - want) <= 2e-6 * max(abs(want), 1e-30)
    print(f"scoring_pbmol on {len(e2e['score'])} described molecules: max rel err {worst:.2e}")
