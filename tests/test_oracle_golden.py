"""The CPU oracle (oracle/pmx_oracle.c) against outputs of the reference itself (tests/golden/, minted by
tests/golden/make_golden.py from /root/reference with the NumPy kernels). This is what pins the oracle."""

import numpy as np
import pytest

from conftest import GOLDEN_SETS, load_golden, rel_err


@pytest.mark.parametrize("name", GOLDEN_SETS)
def test_oracle_reproduces_reference(name, oracle):
    from pharmaconet_amd.constants import weights_vector

    model, lib, weights, d = load_golden(name)
    n = len(lib)
    # set_6oim_c8 holds three trees of 5e6 - 5e7 nodes (the reference needed 14 minutes for them);
    # keep them in: they take the C oracle a few seconds.
    scores, stats = oracle.oracle_score(model.flat, lib, weights_vector(weights), num_threads=8, with_stats=True)
    ref = d["score"]
    assert scores.shape == (n,)
    zero = ref == 0
    assert np.all(scores[zero] == 0.0)
    # the reference's own two kernel variants differ by 3.5e-8; float32 sums, float64 tree
    assert rel_err(scores[~zero], ref[~zero]).max() < 5e-7
    # structure of the search is reproduced exactly
    np.testing.assert_array_equal(stats["n_levels"], d["n_levels"])
    np.testing.assert_array_equal(stats["n_tree"], d["n_tree"])
    np.testing.assert_array_equal(stats["n_leaf"], d["n_leaf"])
    np.testing.assert_array_equal(stats["p_entries"], d["p_entries"])
    np.testing.assert_array_equal(stats["p_invalid"], d["p_invalid"])
    for key in ("s_sum", "p_sum"):
        nz = d[key] != 0
        assert rel_err(stats[key][nz], d[key][nz]).max() < 5e-7


def test_oracle_weight_override_changes_scores(oracle):
    """README.md:172 style override (--hbd 5 --hba 5 --aromatic 8) is honoured (graph_match.py:81-83)."""
    from pharmaconet_amd.constants import weights_vector

    model, lib, weights, d = load_golden("set_6oim_c8_weights")
    assert weights == dict(HBond_donor=5.0, HBond_acceptor=5.0, Aromatic=8.0)
    default = oracle.oracle_score(model.flat, lib, weights_vector(None), num_threads=8)
    assert np.abs(default - d["score"]).max() > 1.0


def test_oracle_thread_count_does_not_change_results(oracle):
    from pharmaconet_amd.constants import weights_vector

    model, lib, weights, d = load_golden("set_6oim_c5")
    a = oracle.oracle_score(model.flat, lib, weights_vector(weights), num_threads=1)
    b = oracle.oracle_score(model.flat, lib, weights_vector(weights), num_threads=4)
    np.testing.assert_array_equal(a, b)


def test_numba_variant_spread(oracle):
    """The reference ships two kernels for the score tables: NumPy (`match_utils.py`, pinned by the fixtures) and Numba
    (`match_utils_numba.py:54-86,126-151`: float64 accumulation, `sigma_sq < 4.0`). A user with Numba installed runs the
    second. Their spread on the fixtures bounds what "within 1e-5 of the reference" has to absorb; a threshold flip
    (2-sigma test, fail count) would show as a jump, so the number of ligands beyond 1e-6 is reported and bounded."""
    from pharmaconet_amd.constants import weights_vector

    worst, flips, total = 0.0, 0, 0
    for name in GOLDEN_SETS:
        model, lib, weights, d = load_golden(name)
        w = weights_vector(weights)
        a = oracle.oracle_score(model.flat, lib, w, num_threads=8, variant="numpy")
        b = oracle.oracle_score(model.flat, lib, w, num_threads=8, variant="numba")
        nz = a != 0
        assert np.all(b[~nz] == 0.0)
        err = rel_err(b[nz], a[nz])
        worst = max(worst, float(err.max()) if err.size else 0.0)
        flips += int((err > 1e-6).sum())
        total += int(nz.sum())
    print(f"numba-variant vs numpy-variant oracle: worst rel diff {worst:.2e}, {flips} of {total} ligands beyond 1e-6")
    assert worst < 1e-5
    assert flips <= total // 100
