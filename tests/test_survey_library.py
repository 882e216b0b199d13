"""SURVEY.md section 8d-2's library generator (tools/survey_library.py): the records are valid packed records, the distributions are the
survey's, a ligand's bytes depend on (seed, ligand index) alone, and the oracle scores them (CPU; the GPU's side is in test_gpu_parity.py)."""

import numpy as np
import pytest

from conftest import load_golden


def _model_nodes(model):
    from pharmaconet_amd.constants import TYPE_ID

    st = model.__getstate__()
    return (np.array([n["center"] for n in st["nodes"]], dtype=np.float64), np.array([TYPE_ID[n["type"]] for n in st["nodes"]]))


@pytest.fixture(scope="module")
def survey():
    from pharmaconet_amd import PackedLibrary
    from tools.survey_library import survey_library

    model, _, _, _ = load_golden("set_6oim_c8")
    centers, types = _model_nodes(model)
    off, data, stats = survey_library(centers, types, 6000, 8, "cpu")
    return model, centers, types, off, data, stats, PackedLibrary(off.numpy().astype(np.uint64), data.numpy())


def test_records_are_valid_and_distributed_as_the_survey_says(survey):
    model, _, _, off, data, stats, lib = survey
    hdr = lib.headers()
    n, c, k = hdr[:, 0].astype(int), hdr[:, 1].astype(int), hdr[:, 2].astype(int)
    assert n.min() >= 4 and n.max() <= 32 and np.all(c == 8) and np.all(k >= 1) and np.all(k <= n)
    assert abs(n.mean() - 20.0) < 0.4 and 5.0 < n.std() < 6.5  # clip(round(N(20, 6)), 4, 32)
    assert abs(stats["active_share"] - 0.1) < 0.02
    want = {"Hydrophobic": 0.45, "HBond_acceptor": 0.20, "HBond_donor": 0.10, "Aromatic": 0.12, "Halogen": 0.05, "Cation": 0.04, "Anion": 0.04}
    for t, share in want.items():
        assert abs(stats["type_share_of_nodes"][t] - share) < 0.04, (t, stats["type_share_of_nodes"])
    raw = data.numpy()
    offs = off.numpy()
    assert np.all(offs % 16 == 0) and np.all(np.diff(offs) > 0)
    for i in range(0, len(lib), 97):  # the structural rules library_stats_kernel enforces, and priority_fn's order of the clusters
        rec = raw[offs[i] : offs[i + 1]]
        ni, ki = int(n[i]), int(k[i])
        tm, ends = rec[8 : 8 + ni], rec[8 + ni : 8 + ni + ki].astype(int)
        assert np.all((tm > 0) & (tm < 128)) and ends[-1] == ni and np.all(np.diff(np.concatenate([[0], ends])) > 0)
        starts = np.concatenate([[0], ends[:-1]])
        group = np.array([0 if tm[s] & 0b0001110 else 1 for s in starts])  # Aromatic / Cation / Anion heads first (graph_match.py:43-60)
        sizes = ends - starts
        key = list(zip(group, -sizes))
        assert key == sorted(key), (i, key)
        xyz = rec[(8 + ni + ki + 3) & ~3 :].view(np.float32)[: ni * 3 * 8]
        assert np.all(np.isfinite(xyz)) and np.abs(xyz).max() < 500.0


def test_a_ligand_depends_on_its_index_alone(survey):
    from tools.survey_library import survey_library

    _, centers, types, off, data, _, _ = survey
    off2, data2, _ = survey_library(centers, types, 700, 8, "cpu", first=5000, chunk=211)  # another shard, another chunking
    a = data[int(off[5000]) : int(off[5700])].numpy()
    assert np.array_equal(a, data2.numpy()) and np.array_equal((off[5000:5701] - off[5000]).numpy(), off2.numpy())
    off3, data3, _ = survey_library(centers, types, 64, 8, "cpu", seed=7)
    assert not np.array_equal(data3.numpy()[:4096], data.numpy()[:4096])


def test_the_oracle_scores_it_and_the_trees_are_not_trivial(survey, oracle):
    from pharmaconet_amd.constants import weights_vector

    model, _, _, _, _, _, lib = survey
    sub = lib.slice(0, 1500)
    sc, st = oracle.oracle_score(model.flat, sub, weights_vector(None), num_threads=8, with_stats=True)
    assert np.all(np.isfinite(sc)) and (sc > 0).mean() > 0.9
    assert st["n_tree"].mean() > 1000 and st["n_levels"].mean() > 5
