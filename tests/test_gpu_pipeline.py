"""`pharmaconet_amd.pipeline.screen_feature_batches`: features -> device packer -> adopted library -> scores -> merged top-k, batch by batch, against one pass over
the host-packed library of all the molecules (the `screening.py:63-70` result)."""

import numpy as np
import pytest

from conftest import load_golden
from test_library import golden_molecules

pytestmark = pytest.mark.gpu


def reference_ranking(model, mols, weights, k):
    from pharmaconet_amd.library import pack_features_native

    lib, status = pack_features_native(mols, threads=4)
    res = model.screen(lib, weights=weights, topk=k)
    return res.topk_indices.cpu().numpy(), res.topk_scores.cpu().numpy(), status


@pytest.mark.parametrize("name", ["set_c21_c8", "set_6oim_c8_weights"])
def test_pipeline_equals_one_pass_over_the_packed_library(name):
    from pharmaconet_amd.library import flatten_features
    from pharmaconet_amd.pipeline import pin_features, screen_feature_batches

    model, lib, weights, d = load_golden(name)
    mols = list(golden_molecules(name))
    cuts = [0, len(mols) // 7, len(mols) // 2, len(mols)]
    batches = [flatten_features(mols[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    want_i, want_s, _ = reference_ranking(model, mols, weights, 25)
    for form in (lambda b: b, pin_features):  # NumPy dicts and pinned tensors
        res = screen_feature_batches(model, (form(b) for b in batches), 25, weights=weights)
        assert res.num_ligands == len(mols) and res.batch_sizes == [b - a for a, b in zip(cuts[:-1], cuts[1:])]
        assert res.num_conformers == 8 * len(mols) and res.num_unsupported == 0 and res.batches_packed_on_host == 0
        assert res.library_bytes == lib.data.size
        np.testing.assert_array_equal(res.topk_indices.cpu().numpy(), want_i)
        np.testing.assert_array_equal(res.topk_scores.cpu().numpy(), want_s)
    assert [i for i, _ in res.ranking()] == want_i.tolist() and res.scores is None
    full = screen_feature_batches(model, batches, 25, weights=weights, keep_scores=True)
    one_pass = model.screen(lib, weights=weights)
    np.testing.assert_array_equal(full.scores.cpu().numpy(), one_pass.scores.cpu().numpy())
    np.testing.assert_array_equal(full.status.cpu().numpy(), one_pass.status.cpu().numpy())


def test_pipeline_hands_a_batch_with_an_oversized_molecule_to_the_host_packer():
    """A molecule beyond the device builder's scratch (300 atoms) in the second batch: that batch is packed on the host, the molecule scored like any other."""
    from pharmaconet_amd.library import LigandFeatures, flatten_features
    from pharmaconet_amd.pipeline import screen_feature_batches

    model, lib, weights, d = load_golden("set_c21_c8")
    mols = list(golden_molecules("set_c21_c8"))[:60]
    na = 300
    ring = [[(i + 1) % na, (i - 1) % na] for i in range(na)]
    rng = np.random.default_rng(4)
    big = LigandFeatures([6] * na, ring, [("Hydrophobic", 5, 5), ("Halogen", 250, 250), ("Cation", 120, 120)], rng.normal(scale=6.0, size=(na, 8, 3)).astype(np.float32))
    too_many = LigandFeatures([17] * 70 + [6], [[70]] * 70 + [list(range(70))], [("Halogen", i, i) for i in range(70)], np.zeros((71, 8, 3), np.float32))  # 70 nodes: unsupported
    mols = mols[:20] + [big] + mols[20:40] + [too_many] + mols[40:]
    batches = [flatten_features(mols[:15]), flatten_features(mols[15:45]), flatten_features(mols[45:])]
    want_i, want_s, status = reference_ranking(model, mols, weights, 30)
    res = screen_feature_batches(model, batches, 30, weights=weights)
    assert res.batches_packed_on_host == 1 and res.num_unsupported == int((status != 0).sum()) == 1
    np.testing.assert_array_equal(res.topk_indices.cpu().numpy(), want_i)
    np.testing.assert_array_equal(res.topk_scores.cpu().numpy(), want_s)


def test_pipeline_without_batches_or_molecules():
    from pharmaconet_amd.library import flatten_features
    from pharmaconet_amd.pipeline import screen_feature_batches

    model, lib, weights, d = load_golden("set_c21_c8")
    res = screen_feature_batches(model, [], 4)
    assert res.num_ligands == 0 and res.topk_indices.cpu().numpy().tolist() == [-1] * 4 and res.ranking() == []
    mols = list(golden_molecules("set_c21_c8"))[:5]
    res = screen_feature_batches(model, [flatten_features([]), flatten_features(mols), flatten_features([])], 3)
    want_i, want_s, _ = reference_ranking(model, mols, None, 3)
    assert res.num_ligands == 5 and res.batch_sizes == [0, 5, 0]
    np.testing.assert_array_equal(res.topk_indices.cpu().numpy(), want_i)
    np.testing.assert_array_equal(res.topk_scores.cpu().numpy(), want_s)
