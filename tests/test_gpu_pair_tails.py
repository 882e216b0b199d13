"""Pair-table entries with one node pair far out in the tails of its tabulated pair function.

`scoring_matching_pair` (match_utils.py:9-74) keeps an entry as long as at most half of its counted node pairs fail the
2-sigma majority test (:61, :71-74) - and the failing pairs still add their likelihood (:63-69). A valid entry can therefore
hold an item 4-7 sigma (or the whole table range) away from every model edge mean; with the CLI's type-weight overrides
(screening.py:54-62) such an item can belong to a pair function whose weights are a million times those of the items that
pass. These tests sweep that regime: for model cluster pairs (a, b) of the three fixture models, reduced to those two
clusters, two-cluster ligands whose pair entry has L1 x L2 = 2 or 4 node pairs - one swept from 0 to the table range + 5 A
(or as far as the triangle inequalities let it go), the others pinned where their own function passes the majority test -
under the default weights and under overrides with weight ratios up to 1000 (w x w ratio 10^6), GPU against the oracle.
"""

import dataclasses
import itertools
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

RTOL = 2e-6  # where the oracle's score is above FLOOR
FLOOR = 1e-30

WEIGHT_SETS = {
    "default": None,
    "charged_heavy": {"Cation": 100.0, "Anion": 100.0, "Hydrophobic": 0.1},  # VERDICT r4's example (--cation 100 --hydrophobic 0.1)
    "charged_light": {"Cation": 0.1, "Anion": 0.1, "Hydrophobic": 100.0},   # ... and its inverse
    # types that share clusters in the fixture models (Cation + HBond, Aromatic + Hydrophobic) a factor 1000 apart
    "core_heavy": {"Cation": 100.0, "Anion": 100.0, "Aromatic": 100.0, "HBond_donor": 0.1, "HBond_acceptor": 0.1, "Hydrophobic": 0.1, "Halogen": 0.1},
    "core_light": {"Cation": 0.1, "Anion": 0.1, "Aromatic": 0.1, "HBond_donor": 100.0, "HBond_acceptor": 100.0, "Hydrophobic": 100.0, "Halogen": 100.0},
}


class _Shim:
    """What engine.screen needs of a model: `.flat` and a slot for the device handle."""

    def __init__(self, flat):
        self.flat = flat
        self._engine_handle = None


def _record(masks_a, masks_b, xyz: np.ndarray) -> bytes:
    """Two ligand clusters (nodes of `masks_a`, then of `masks_b`); xyz float32 [n][3][C]."""
    n, C = xyz.shape[0], xyz.shape[2]
    head = struct.pack("<HHHH", n, C, 2, 0) + bytes(list(masks_a) + list(masks_b)) + bytes([len(masks_a), n])
    head += b"\0" * (-len(head) % 4)
    rec = head + np.ascontiguousarray(xyz, dtype=np.float32).tobytes()
    return rec + b"\0" * (-len(rec) % 16)


def _subsets(flat, a: int):
    """Distinct non-empty node subsets of model cluster `a` by ligand type mask (graph_match.py:148-150): {node tuple: mask}."""
    nodes = int(np.asarray(flat.cluster_nodes).reshape(flat.num_clusters, -1)[a, 0])
    subs = {}
    for mask in range(1, 128):
        sub = tuple(m for m in range(flat.num_nodes) if (nodes >> m) & 1 and (mask >> int(flat.node_type[m])) & 1)
        if sub and sub not in subs:
            subs[sub] = mask
    return subs


def _two_cluster_model(flat, a: int, b: int):
    """The model reduced to clusters a and b (all nodes and edges kept, so node indices stand)."""
    idx = [a, b]
    return dataclasses.replace(
        flat,
        cluster_nodes=np.asarray(flat.cluster_nodes)[idx].copy(),
        cluster_typemask=flat.cluster_typemask[idx].copy(),
        cluster_center=flat.cluster_center[idx].copy(),
        cluster_size=flat.cluster_size[idx].copy(),
        cluster_type=tuple(flat.cluster_type[i] for i in idx),
    )


def _pin(flat, A, B):
    """A distance at which the item of node subsets (A, B) passes the majority test of match_utils.py:55-61 (the edge mean that
    most terms lie within 2 sigma of), or None."""
    mean = flat.edge_mean.astype(np.float32)
    std = flat.edge_std.astype(np.float32)
    best, best_d = -1, None
    for m, n in itertools.product(A, B):
        d = np.float32(mean[m, n])
        if not np.isfinite(d):
            continue
        cnt = sum(1 for mm, nn in itertools.product(A, B) if abs(np.float32(d - mean[mm, nn]) / std[mm, nn]) < 2.0)
        if cnt > best:
            best, best_d = cnt, float(d)
    if best_d is None or 2 * best < len(A) * len(B):
        return None
    return best_d


def _coords_12(t, p03):
    """Nodes 0 | 2, 3: d02 = t (swept), d03 = p03. Returns [3 nodes][3][C]."""
    C = len(t)
    xyz = np.zeros((3, 3, C))
    xyz[1, 0, :] = t
    xyz[2, 1, :] = p03
    return xyz


def _coords_22(t, p03, p12, p13):
    """Nodes 0, 1 | 2, 3 with d02 = t (swept), d03, d12, d13 pinned; None where the triangle inequalities rule t out."""
    t = np.asarray(t, dtype=np.float64)
    lo = np.maximum(np.abs(t - p12), abs(p03 - p13))
    hi = np.minimum(t + p12, p03 + p13)
    if np.any(lo > hi - 1e-6):
        return None
    D = 0.5 * (lo + hi)  # d01
    cos_th = np.clip((p03 * p03 + p13 * p13 - D * D) / (2 * p03 * p13), -1.0, 1.0)
    C = len(t)
    xyz = np.zeros((4, 3, C))
    xyz[0, 0, :] = p03                                   # node 0 (node 3 at the origin)
    xyz[1, 0, :] = p13 * cos_th
    xyz[1, 1, :] = p13 * np.sqrt(1.0 - cos_th * cos_th)  # node 1
    e = (xyz[1] - xyz[0]) / D
    al = (t * t - p12 * p12 + D * D) / (2 * D)
    h = np.sqrt(np.maximum(t * t - al * al, 0.0))
    xyz[2] = xyz[0] + al * e
    xyz[2, 2, :] += h                                    # node 2
    return xyz


def _combos(subs_a, subs_b, weights7, node_type, limit, rng):
    """Mask combinations for the two shapes, those with the largest weight contrast between the swept and the pinned items first."""
    def wt(sub):
        return max(weights7[int(node_type[m])] for m in sub)

    A, B = list(subs_a.items()), list(subs_b.items())
    out12, out22 = [], []
    for (sa, ma) in A:
        for (sb1, mb1), (sb2, mb2) in itertools.product(B, B):
            out12.append((abs(np.log(wt(sb1) / wt(sb2))), (ma,), (mb1, mb2), (sa,), (sb1, sb2)))
    for (sa1, ma1), (sa2, ma2) in itertools.product(A, A):
        for (sb1, mb1), (sb2, mb2) in itertools.product(B, B):
            hi = wt(sa1) * wt(sb1)
            lo = min(wt(sa1) * wt(sb2), wt(sa2) * wt(sb1), wt(sa2) * wt(sb2))
            out22.append((abs(np.log(hi / lo)), (ma1, ma2), (mb1, mb2), (sa1, sa2), (sb1, sb2)))
    picked = []
    for out in (out12, out22):
        rng.shuffle(out)
        out.sort(key=lambda x: -x[0])
        picked += out[: max(1, limit // 2)] + out[len(out) // 2 : len(out) // 2 + max(1, limit // 4)]
    return picked


def _library_for_pair(flat, a, b, weights7, dmax, delta, n_conf, limit, rng):
    from pharmaconet_amd import PackedLibrary

    subs_a, subs_b = _subsets(flat, a), _subsets(flat, b)
    recs = []
    for _, ma, mb, sa, sb in _combos(subs_a, subs_b, weights7, flat.node_type, limit, rng):
        if len(ma) == 1:
            p03 = _pin(flat, sa[0], sb[1])
            if p03 is None:
                continue
            tmax = dmax
        else:
            pins = [_pin(flat, sa[0], sb[1]), _pin(flat, sa[1], sb[0]), _pin(flat, sa[1], sb[1])]
            if any(p is None for p in pins):
                continue
            p03, p12, p13 = pins
            tmax = min(dmax, p03 + p12 + p13 - 0.05)
        per = int(np.ceil(tmax / (delta * n_conf)))
        for j in range(per):
            t = (j * n_conf + np.arange(n_conf) + 1) * delta
            t = np.minimum(t, tmax)
            xyz = _coords_12(t, p03) if len(ma) == 1 else _coords_22(t, p03, p12, p13)
            if xyz is None:
                continue
            recs.append(_record(ma, mb, xyz.astype(np.float32)))
    return PackedLibrary.from_records(recs) if recs else None


@pytest.mark.parametrize("name,max_pairs,delta", [("model_6oim_like", 10 ** 6, 0.06), ("model_clustered21", 10 ** 6, 0.07), ("model_stress64", 160, 0.08)])
def test_pair_entries_with_one_item_in_the_tails_match_the_oracle(name, max_pairs, delta, oracle):
    from pharmaconet_amd import PharmacophoreModel, engine
    from pharmaconet_amd.constants import weights_vector

    flat = PharmacophoreModel.load(GOLDEN / f"{name}.pm").flat
    dmax = float(np.nanmax(flat.edge_mean.astype(np.float64) + 7.0 * flat.edge_std.astype(np.float64))) + 5.0
    rng = np.random.default_rng(20250929)
    pairs = [(a, b) for a in range(flat.num_clusters) for b in range(a + 1, flat.num_clusters)]
    # The suite takes a seeded sample of the cluster pairs (the whole sweep - 55 + 153 + 160 pairs, four minutes with the oracle on 256 host
    # threads - runs with PMX_FULL_SWEEPS=1 and is what DESIGN.md section 5 quotes).
    if not os.environ.get("PMX_FULL_SWEEPS"):
        max_pairs = min(max_pairs, 36)
    n_all = len(pairs)
    if len(pairs) > max_pairs:
        pairs = [pairs[i] for i in sorted(rng.choice(len(pairs), size=max_pairs, replace=False))]
    threads = os.cpu_count() or 8
    acc = {w: dict(worst=0.0, at=None, n_scores=0, n_valid=0, n_exactv=0) for w in WEIGHT_SETS}
    for a, b in pairs:
        sub = _two_cluster_model(flat, a, b)
        shim = _Shim(sub)  # one device model per cluster pair, every weight set against it
        for wname, wdict in WEIGHT_SETS.items():
            w7 = weights_vector(wdict)
            st = acc[wname]
            lib = _library_for_pair(flat, a, b, w7, dmax, delta, 8, 8, rng)
            if lib is None:
                continue
            ref, stats = oracle.oracle_score(sub, lib, w7, num_threads=threads, with_stats=True)
            res = engine.screen(shim, lib, weights=wdict)
            got = res.scores.cpu().numpy().astype(np.float64)
            assert np.all(res.status.cpu().numpy() == 0)
            st["n_exactv"] += engine.last_score_stats()["n_exact_values"]
            sel = ref > FLOOR
            assert np.all(got[~sel] <= FLOOR * (1 + RTOL)), "scores the oracle puts below the floor"
            st["n_valid"] += int((stats["p_entries"] - stats["p_invalid"] > 0).sum())
            if sel.any():
                err = rel_err(got[sel], ref[sel])
                st["n_scores"] += int(sel.sum())
                if err.max() > st["worst"]:
                    st["worst"], st["at"] = float(err.max()), (a, b, float(ref[sel][err.argmax()]))
    for wname, st in acc.items():
        print(f"{name} [{wname}]: {len(pairs)} cluster pairs, {st['n_scores']} scores above {FLOOR:g}, {st['n_valid']} ligands with a valid pair entry, "
              f"{st['n_exactv']} items term by term; max rel err {st['worst']:.2e} at {st['at']}")
    for wname, st in acc.items():
        assert st["n_valid"] >= 1000 * len(pairs) // n_all, f"[{wname}] the sweep does not reach valid pair entries"
        assert st["worst"] <= RTOL, f"[{wname}] max rel err {st['worst']:.3e} at (a, b, oracle score) = {st['at']}"


def _heavy_light(flat, a, w7):
    """The node subsets of cluster `a` made of its heaviest-weighted type only and of its lightest only: ((nodes, mask), (nodes, mask)) or None."""
    subs = _subsets(flat, a)
    ws = sorted({w7[int(flat.node_type[m])] for sub in subs for m in sub})
    if len(ws) < 2:
        return None
    pick = {}
    for sub, mask in subs.items():
        kinds = {w7[int(flat.node_type[m])] for m in sub}
        for name, w in (("H", ws[-1]), ("L", ws[0])):
            if kinds == {w} and (name not in pick or len(sub) > len(pick[name][0])):
                pick[name] = (sub, mask)
    return (pick["H"], pick["L"]) if len(pick) == 2 else None


def _rescaled(flat, sigma_min_to: float):
    """The same model with every length multiplied so that its smallest edge sigma is `sigma_min_to` - just above a power of two, the
    tabulated functions' grid (h = the largest power of two <= sigma_min / 4) is as coarse against the sigmas as it can get."""
    f = np.float64(sigma_min_to) / np.float64(np.nanmin(flat.edge_std))
    return dataclasses.replace(
        flat,
        edge_mean=(flat.edge_mean.astype(np.float64) * f).astype(np.float32),
        edge_std=(flat.edge_std.astype(np.float64) * f).astype(np.float32),
        cluster_center=flat.cluster_center * f,
        cluster_size=flat.cluster_size * f,
    )


@pytest.mark.parametrize("wname,sigma_min_to", [("core_heavy", None), ("core_light", None), ("charged_heavy", None), ("core_heavy", 1.0000005), ("core_light", 2.000001)])
def test_one_heavy_item_in_the_tails_beside_light_items_that_pass(wname, sigma_min_to, oracle, monkeypatch):
    """The entry the 2- and 4-pair sweeps above cannot build (their mixed items cap the contrast at w_max / w_min): two ligand
    clusters of three light-typed nodes and one heavy-typed node each. The nine light x light node pairs sit where their function
    passes, the six heavy x light ones 25 A away (they fail and add nothing), and the ONE heavy x heavy pair is swept through the
    tails of its function: 7 of 16 counted pairs fail, the entry stands (match_utils.py:71-74) - and between 3 and 6 sigma it is
    little else than that one tail value, w_max^2 / w_min^2 = 10^6 times the weight of the items that pass. Pair items evaluate
    rough cells term by term when the call's weights are this far apart (item_finish<TAILS>); PMX_PAIR_TAILS=0 shows what the
    tabulated value alone would have given (the fixture model's sigmas are 5.6 grid steps and more; rescaled so that the smallest is
    exactly 4, the tabulated tail alone is off by 10^-5)."""
    from pharmaconet_amd import PackedLibrary, PharmacophoreModel, engine
    from pharmaconet_amd.constants import weights_vector

    flat = PharmacophoreModel.load(GOLDEN / "model_6oim_like.pm").flat
    if sigma_min_to is not None:
        flat = _rescaled(flat, sigma_min_to)
    wdict = WEIGHT_SETS[wname]
    w7 = weights_vector(wdict)
    rng = np.random.default_rng(7)
    far = 12.0 * float(np.nanmax(flat.edge_std))  # where the heavy nodes sit, off the axis of the light ones
    step = 0.0125 * float(np.nanmin(flat.edge_std)) / 1.4
    worst, worst_off, n_scores, n_tail, n_pairs, n_exactv = 0.0, 0.0, 0, 0, 0, 0
    for a in range(flat.num_clusters):
        for b in range(flat.num_clusters):
            if a == b:
                continue
            ha, hb = _heavy_light(flat, a, w7), _heavy_light(flat, b, w7)
            if ha is None or hb is None:
                continue
            (Ha, mHa), (La, mLa) = ha
            (Hb, mHb), (Lb, mLb) = hb
            p_ll, p_hh = _pin(flat, La, Lb), _pin(flat, Ha, Hb)
            if p_ll is None or p_hh is None:
                continue
            n_pairs += 1
            jit = rng.normal(size=(2, 3, 3)) * 0.14 * float(np.nanmin(flat.edge_std))
            recs = []
            for j in range(200):
                t = np.maximum(p_hh - 240 * step + (j * 8 + np.arange(8)) * step, 0.05)  # the heavy pair, from 2 sigma inside its mean to 10 sigma beyond
                xyz = np.zeros((8, 3, 8))
                for k in range(3):
                    xyz[k] = jit[0, k][:, None]
                    xyz[4 + k] = jit[1, k][:, None]
                    xyz[4 + k, 0, :] += p_ll
                xyz[3, 0, :], xyz[3, 1, :] = p_ll / 2 - t / 2, far
                xyz[7, 0, :], xyz[7, 1, :] = p_ll / 2 + t / 2, far
                recs.append(_record([mLa] * 3 + [mHa], [mLb] * 3 + [mHb], xyz.astype(np.float32)))
            lib = PackedLibrary.from_records(recs)
            sub = _two_cluster_model(flat, a, b)
            ref, stats = oracle.oracle_score(sub, lib, w7, num_threads=os.cpu_count() or 8, with_stats=True)
            assert np.all(stats["p_entries"] - stats["p_invalid"] > 0)
            floor_score = ref.min()  # the far end of the sweep: the light items alone
            n_tail += int(((ref > 1.05 * floor_score) & (ref < 50 * floor_score)).sum())  # entries that are mostly the heavy tail item
            shim = _Shim(sub)
            got = engine.screen(shim, lib, weights=wdict).scores.cpu().numpy().astype(np.float64)
            n_exactv += engine.last_score_stats()["n_exact_values"]
            worst = max(worst, float(rel_err(got, ref).max()))
            n_scores += len(ref)
            monkeypatch.setenv("PMX_PAIR_TAILS", "0")
            off = engine.screen(shim, lib, weights=wdict).scores.cpu().numpy().astype(np.float64)
            monkeypatch.delenv("PMX_PAIR_TAILS")
            worst_off = max(worst_off, float(rel_err(off, ref).max()))
    print(f"[{wname}, sigma_min {'as is' if sigma_min_to is None else sigma_min_to}] {n_pairs} cluster pairs, {n_scores} scores, {n_tail} of them mostly one tail item; {n_exactv} items term by term; "
          f"max rel err {worst:.2e} (tabulated tails alone, PMX_PAIR_TAILS=0: {worst_off:.2e})")
    assert n_pairs >= 6 and n_tail >= 100
    assert worst <= RTOL
