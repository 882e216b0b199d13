"""Self-table entries far from the peaks of the tabulated pair functions.

`scoring_matching_self` (match_utils.py:77-122) has no pass logic: a ligand whose best leaf is one matched multi-node
cluster scores its self entry alone, and that entry can lie anywhere in the tails of the Gaussians - down to nothing. The
engine tabulates the sums over model node pairs as functions of the one distance; where a cell of the table is not accurate
*relative to the function's own value* the self loop evaluates the terms one by one. These tests sweep that: for every
cluster of the fixture models and every pair of its node subsets, two-node single-cluster ligands whose node distance runs
from 0 to 5 A beyond the table's range, against the oracle."""

import dataclasses
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

RTOL = 2e-6  # where the oracle's score is above 1e-30
FLOOR = 1e-30


class _Shim:
    """What engine.screen needs of a model: `.flat` and a slot for the device handle."""

    def __init__(self, flat):
        self.flat = flat
        self._engine_handle = None


def _two_node_record(mask1: int, mask2: int, d: np.ndarray) -> bytes:
    """One cluster of two nodes, node 0 at the origin and node 1 at (d[c], 0, 0) in conformer c."""
    C = len(d)
    xyz = np.zeros((2, 3, C), dtype=np.float32)
    xyz[1, 0, :] = d
    head = struct.pack("<HHHH", 2, C, 1, 0) + bytes([mask1, mask2]) + bytes([2])
    head += b"\0" * (-len(head) % 4)
    rec = head + xyz.tobytes()
    return rec + b"\0" * (-len(rec) % 16)


def _subset_pairs(flat, a: int):
    """Distinct non-empty node subsets of model cluster `a` by ligand type mask (graph_match.py:148-150), one mask each, and
    every unordered pair of them (the same subset twice included: two ligand nodes of one type)."""
    nodes = int(flat.cluster_nodes[a])
    subs = {}
    for mask in range(1, 128):
        sub = 0
        for m in range(flat.num_nodes):
            if (nodes >> m) & 1 and (mask >> int(flat.node_type[m])) & 1:
                sub |= 1 << m
        if sub and sub not in subs:
            subs[sub] = mask
    masks = list(subs.values())
    return [(masks[i], masks[j]) for i in range(len(masks)) for j in range(i, len(masks))]


def _single_cluster_model(flat, a: int):
    """The model reduced to cluster `a` (all nodes and edges kept, so node indices stand): the two-node ligand then has one
    level with one candidate, and its score is the mean over conformers of max(0, S[0][a])."""
    return dataclasses.replace(
        flat,
        cluster_nodes=flat.cluster_nodes[a:a + 1].copy(),
        cluster_typemask=flat.cluster_typemask[a:a + 1].copy(),
        cluster_center=flat.cluster_center[a:a + 1].copy(),
        cluster_size=flat.cluster_size[a:a + 1].copy(),
        cluster_type=tuple(flat.cluster_type[a:a + 1]),
    )


def _sweep_library(pairs, dmax: float, n_conf: int, delta: float):
    """For every mask pair, ligands whose conformers step through [0, dmax] in `delta` increments."""
    from pharmaconet_amd import PackedLibrary

    per = int(np.ceil(dmax / (delta * n_conf)))
    recs, dist = [], []
    for m1, m2 in pairs:
        for j in range(per):
            d = ((j * n_conf + np.arange(n_conf)) * delta).astype(np.float32)
            recs.append(_two_node_record(m1, m2, d))
            dist.append(d)
    return PackedLibrary.from_records(recs), np.array(dist)


def _gpu(model, lib, monkeypatch=None, flags=None):
    from pharmaconet_amd import engine

    if flags is not None:
        monkeypatch.setenv("PMX_TREE_FLAGS", str(flags))
    res = engine.screen(model, lib)
    out = res.scores.cpu().numpy().astype(np.float64), res.status.cpu().numpy()
    stats = engine.last_score_stats()
    if flags is not None:
        monkeypatch.delenv("PMX_TREE_FLAGS")
    return out + (stats,)


def _table_range(flat) -> float:
    return float(np.nanmax(flat.edge_mean.astype(np.float64) + 7.0 * flat.edge_std.astype(np.float64)))


@pytest.mark.parametrize("name,n_conf,delta", [("model_6oim_like", 8, 0.011), ("model_6oim_like", 1, 0.0173),
                                               ("model_clustered21", 8, 0.013), ("model_stress64", 8, 0.017)])
def test_self_entries_in_the_tails_match_the_oracle(name, n_conf, delta, oracle, monkeypatch):
    from pharmaconet_amd import PharmacophoreModel
    from pharmaconet_amd.constants import weights_vector

    flat = PharmacophoreModel.load(GOLDEN / f"{name}.pm").flat
    dmax = _table_range(flat) + 5.0
    worst, worst_vs_exact, n_checked, n_slow, n_pairs = 0.0, 0.0, 0, 0, 0
    for a in range(flat.num_clusters):
        pairs = _subset_pairs(flat, a)
        if not pairs:
            continue
        n_pairs += len(pairs)
        sub = _single_cluster_model(flat, a)
        lib, _ = _sweep_library(pairs, dmax, n_conf, delta)
        ref = oracle.oracle_score(sub, lib, weights_vector(None), num_threads=os.cpu_count() or 8)
        got, status, stats = _gpu(_Shim(sub), lib)
        n_slow += stats["n_exact_values"]
        assert np.all(status == 0)
        sel = ref > FLOOR
        assert np.all(got[~sel] <= FLOOR * (1 + RTOL)), "scores the oracle puts below the floor"
        err = rel_err(got[sel], ref[sel])
        worst = max(worst, float(err.max()))
        n_checked += int(sel.sum())
        assert err.max() <= RTOL, (f"cluster {a}: max rel err {err.max():.3e} at oracle score {ref[sel][err.argmax()]:.3e}")
        # the term-by-term table phase (PMX_TREE_FLAGS=8) against the default one
        exact, _, _ = _gpu(_Shim(sub), lib, monkeypatch, flags=8)
        dev = rel_err(got[sel], exact[sel])
        worst_vs_exact = max(worst_vs_exact, float(dev.max()))
        assert rel_err(exact[sel], ref[sel]).max() <= RTOL
    print(f"{name} C={n_conf}: {n_pairs} subset pairs, {n_checked} scores above {FLOOR:g}; max rel err vs oracle {worst:.2e}, "
          f"default vs term-by-term {worst_vs_exact:.2e}; {n_slow} self items evaluated term by term")
    assert n_slow > 0  # the sweep reaches the flagged cells
    # The term-by-term phase reproduces the reference's float32 rounding of z and z^2 (up to 1.8e-7 z^2 / 2 of a term), a table of
    # the smooth function cannot: cells are evaluated term by term from exponent 6 on, below that the two differ by up to
    # 6 * 1.8e-7 + the table's 2e-7 + the float32 Horner steps.
    assert worst_vs_exact <= 1.8e-6


def test_tail_self_entry_decides_a_whole_ligand(oracle):
    """The adversarial ligand itself, against the full model: one ring-like cluster of three nodes whose distances sit 4.5 - 9
    sigma from every model edge mean they can be matched with, plus single-node clusters that match nothing (so the best leaf is
    the self entry alone)."""
    from pharmaconet_amd import PackedLibrary, PharmacophoreModel
    from pharmaconet_amd.constants import TYPE_ID, weights_vector

    model = PharmacophoreModel.load(GOLDEN / "model_6oim_like.pm")
    flat = model.flat
    rng = np.random.default_rng(424242)
    recs = []
    for _ in range(512):
        C = 8
        t = int(rng.choice([TYPE_ID["Aromatic"], TYPE_ID["Hydrophobic"], TYPE_ID["Cation"], TYPE_ID["HBond_donor"]]))
        scale = rng.uniform(9.0, 24.0)  # node distances well beyond the model's intra-cluster means
        pts = rng.normal(size=(3, 3)) * scale
        xyz = np.zeros((3, 3, C), dtype=np.float32)
        for c in range(C):
            xyz[:, :, c] = pts + rng.normal(size=(3, 3)) * 0.05
        head = struct.pack("<HHHH", 3, C, 1, 0) + bytes([1 << t] * 3) + bytes([3])
        head += b"\0" * (-len(head) % 4)
        rec = head + xyz.tobytes()
        recs.append(rec + b"\0" * (-len(rec) % 16))
    lib = PackedLibrary.from_records(recs)
    from pharmaconet_amd import engine

    ref = oracle.oracle_score(flat, lib, weights_vector(None), num_threads=os.cpu_count() or 8)
    res = engine.screen(model, lib)
    got = res.scores.cpu().numpy().astype(np.float64)
    sel = ref > FLOOR
    tiny = sel & (ref < 1e-4)
    print(f"{int(sel.sum())} ligands above the floor, {int(tiny.sum())} with scores below 1e-4; "
          f"max rel err {rel_err(got[sel], ref[sel]).max():.2e}")
    assert tiny.sum() >= 32
    assert rel_err(got[sel], ref[sel]).max() <= RTOL
    assert np.all(got[~sel] <= FLOOR * (1 + RTOL))
