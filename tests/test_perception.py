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


def test_native_perception_of_the_whole_batch_matches_the_reference(golden):
    """`pmx_perceive_features` (csrc/pmx_perceive.cpp) on all 600 described molecules in ONE call - the toolkit's per-atom answers in,
    flat feature lists out - against the reference's own `get_pharmacophore_nodes`; and chained into `pmx_pack_features`, the
    records of the reference's LigandGraph. One thread and many give the same arrays."""
    from pharmaconet_amd.library import pack_features_native
    from pharmaconet_amd.ligand import features_of, perceive_batch, toolkit_answers

    answers, positions = [], []
    for desc in golden["molecules"]:
        pb = fake_openbabel.Molecule(desc)
        pb.removeh()
        answers.append(toolkit_answers(pb))
        positions.append(np.ascontiguousarray(np.moveaxis(np.asarray(desc["coords"], dtype=np.float32), 0, 1)))  # [atoms, conformers, 3]
    flat = perceive_batch(answers, positions, threads=1)
    many = perceive_batch(answers, positions, threads=8)
    assert all(np.array_equal(flat[k], many[k]) for k in flat)
    for i, want in enumerate(golden["reference"]):
        got = [[t, _plain(a), _plain(c)] for t, a, c in features_of(flat, i)]
        assert got == want, f"molecule {i}"
    # ... and on into the packer without a Python loop: records equal to the reference's LigandGraph (perception_e2e.npz holds the first 96)
    lib, status = pack_features_native(flat, threads=4)
    e2e = np.load(GOLDEN / "perception_e2e.npz")
    off = np.concatenate([[0], np.cumsum(e2e["record_len"])])
    for i in range(len(e2e["record_len"])):
        assert status[i] == 0
        assert lib.record(i) == e2e["records"][off[i]:off[i + 1]].tobytes(), f"molecule {i}"


def test_malformed_toolkit_answers_are_reported(golden):
    from pharmaconet_amd.ligand import perceive_batch, toolkit_answers

    pb = fake_openbabel.Molecule(golden["molecules"][0])
    pb.removeh()
    ans = toolkit_answers(pb)
    ans["nbr"] = ans["nbr"].copy()
    ans["nbr"][0] = 10_000  # a neighbour outside the molecule
    with pytest.raises(ValueError):
        perceive_batch([ans])


def test_screening_reads_a_directory_in_one_batch(golden, tmp_path, monkeypatch):
    """`screening.load_library` on a directory (screening.py:63-68): files are read in worker processes, perception rules and
    packing run once on the batch - the library equals the per-molecule path's."""
    from pharmaconet_amd import screening
    from pharmaconet_amd.library import pack_ligand
    from pharmaconet_amd.ligand import Ligand

    picks = [0, 3, 5, 8, 13]
    for i in picks:
        (tmp_path / f"mol{i:03d}.sdf").write_text(json.dumps(golden["molecules"][i]))

    class _Pool:  # (the stand-in toolkit lives in this process only)
        def __init__(self, n):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def map(self, fn, items):
            return [fn(x) for x in items]

    monkeypatch.setattr(screening.multiprocessing, "Pool", _Pool)
    names, lib = screening.load_library(tmp_path, cpus=2)
    assert len(names) == len(picks) == len(lib)
    for k, name in enumerate(names):
        i = int(name[-7:-4])
        desc = golden["molecules"][i]
        one = Ligand(fake_openbabel.Molecule(desc), np.asarray(desc["coords"], dtype=np.float32), conformer_axis=0)
        assert lib.record(k) == bytes(pack_ligand(one.features))


def test_native_rules_equal_the_python_restatement_on_random_molecules(golden):
    """4 000 random molecular graphs with random toolkit answers (elements, bonds, hybridisation, acceptor / donor flags, ring sets -
    chemically meaningless on purpose: every rule sees neighbourhoods the fixture generator never drew): the native rules
    (`pmx_perceive_features`) against the Python restatement that the 600 reference outputs pin as well (tests/perception_rules.py)."""
    import perception_rules

    from pharmaconet_amd.ligand import features_of, perceive_batch, toolkit_answers

    # (the restatement is itself held to the reference first)
    for desc, want in list(zip(golden["molecules"], golden["reference"]))[:100]:
        pb = fake_openbabel.Molecule(desc)
        pb.removeh()
        assert [[t, _plain(a), _plain(c)] for t, a, c in perception_rules.perceive_features(pb)[2]] == want
    rng = np.random.default_rng(20250930)
    elements = np.array([6, 6, 6, 6, 7, 7, 8, 8, 9, 15, 16, 17, 35, 53])
    descs = []
    for _ in range(4000):
        n = int(rng.integers(1, 24))
        z = rng.choice(elements, size=n).tolist()
        bonds = set()
        for i in range(1, n):  # a random tree, then a few ring closures
            if rng.random() < 0.9:
                bonds.add((int(rng.integers(0, i)), i))
        for _ in range(int(rng.integers(0, 4))):
            a, b = sorted(rng.integers(0, n, size=2).tolist())
            if a != b:
                bonds.add((a, b))
        order = list(bonds)
        rng.shuffle(order)
        rings = []
        for _ in range(int(rng.integers(0, 3))):
            if n >= 3:
                ring = rng.choice(n, size=int(rng.integers(3, min(n, 7) + 1)), replace=False).tolist()
                rings.append([ring, bool(rng.random() < 0.6)])
        descs.append(dict(z=z, bonds=[list(b) if rng.random() < 0.5 else [b[1], b[0]] for b in order],
                          acceptor=(rng.random(n) < 0.3).tolist(), donor=(rng.random(n) < 0.25).tolist(),
                          hyb=rng.integers(1, 4, size=n).tolist(), rings=rings, rotors=0, coords=[np.zeros((n, 3)).tolist()]))
    answers, want = [], []
    for desc in descs:
        pb = fake_openbabel.Molecule(desc)
        pb.removeh()
        want.append(perception_rules.perceive_features(pb)[2])
        answers.append(toolkit_answers(pb))
    flat = perceive_batch(answers, threads=4)
    n_feat = 0
    for i, w in enumerate(want):
        assert features_of(flat, i) == w, f"molecule {i}: {descs[i]}"
        n_feat += len(w)
    kinds = {t for w in want for t, _, _ in w}
    assert kinds == {"Hydrophobic", "Aromatic", "Cation", "Anion", "HBond_donor", "HBond_acceptor", "Halogen"} and n_feat > 20000
