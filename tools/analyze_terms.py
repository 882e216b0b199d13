#!/usr/bin/env python3
"""Work accounting of the pair-score tables (match_utils.py:9-122) on the bench library, CPU only.

Per ligand-conformer it counts
  direct      Gaussian terms the reference evaluates: sum over near entries (a, b), ligand node pairs (u, v) of |A||B|
  unique      distinct (u, v, m, n) among them (a model node sits in several clusters: density_map.py:131-177)
  unique_all  distinct (u, v, m, n) without the cluster-distance prefilter (graph_match.py:263-268)
  direct_all  terms without the prefilter
and the sizes that decide a lane mapping (entries, near entries, items, node-list lengths); then it replays
tables_kernel_v2's mapping (8 items side by side, columns in batches of W) and prints the share of term slots that
carry a term, for W = 1, 2, 3 (the kernel uses 3).

    python tools/analyze_terms.py [--ligands 400] [--model tests/golden/model_6oim_like.pm]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ligands", type=int, default=400)
    ap.add_argument("--model", default=str(REPO / "tests/golden/model_6oim_like.pm"))
    ap.add_argument("--conformers", type=int, default=8)
    args = ap.parse_args()
    from pharmaconet_amd import PharmacophoreModel
    from pharmaconet_amd.constants import TYPE_ID
    from pharmaconet_amd.synthetic import BASE_SEED, synthetic_library

    model = PharmacophoreModel.load(args.model)
    fm = model.flat
    Nm, K = fm.num_nodes, fm.num_clusters
    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    lib = synthetic_library(args.ligands, num_conformers=args.conformers, model_nodes=(centers, types), active_fraction=0.1,
                            seed=BASE_SEED, max_nodes=32, conformer_noise=0.45)
    cn = [int(x) for x in fm.cluster_nodes]
    memb = sum(bin(x).count("1") for x in cn)
    union = 0
    for x in cn:
        union |= x
    print(f"model: {Nm} nodes, {K} clusters, sizes {[bin(x).count('1') for x in cn]}, memberships {memb}, nodes in clusters {bin(union).count('1')}")
    print("cluster types:", fm.cluster_type)
    tnodes = np.zeros(128, dtype=object)
    for mask in range(128):
        v = 0
        for m in range(Nm):
            if (mask >> int(fm.node_type[m])) & 1:
                v |= 1 << m
        tnodes[mask] = v
    cc = fm.cluster_center
    cdist = np.sqrt(((cc[:, None, :] - cc[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    csize = (fm.cluster_size[:, None] + fm.cluster_size[None, :]).astype(np.float32)

    tot = dict(direct=0, unique=0, unique_all=0, direct_all=0, entries=0, near=0, items=0, items_near=0, levels=0,
               lig_pairs=0, self_direct=0, self_unique=0, rowsum_adds=0, entry_adds=0)
    hist_listlen = np.zeros(65, dtype=np.int64)
    for li in range(len(lib)):
        r = lib.unpack(li)
        n, C = r["n_nodes"], r["n_conf"]
        tm = r["typemask"]
        ends = r["cluster_end"]
        xyz = r["xyz"]  # [n,3,C]
        levels = []
        start = 0
        for ci in range(r["n_clusters"]):
            end = int(ends[ci])
            lmask = 0
            for u in range(start, end):
                lmask |= int(tm[u])
            cand = [a for a in range(K) if int(fm.cluster_typemask[a]) & lmask]
            if cand and len(levels) < 20:
                levels.append((start, end, cand))
            start = end
        nl = len(levels)
        tot["levels"] += nl
        # cluster geometry per conformer
        geo = []
        for (s, e, cand) in levels:
            p = xyz[s:e].astype(np.float32)  # [k,3,C]
            ctr = p.sum(0) / np.float32(e - s)
            size = np.sqrt(((p - ctr[None]) ** 2).sum(1)).max(0)
            geo.append((ctr, size))
        for i in range(nl):
            si, ei, ci_ = levels[i]
            # self tables
            for a in ci_:
                for u in range(si, ei):
                    for v in range(u + 1, ei):
                        A = cn[a] & tnodes[int(tm[u])]
                        B = cn[a] & tnodes[int(tm[v])]
                        tot["self_direct"] += bin(A).count("1") * bin(B).count("1")
            for u in range(si, ei):
                for v in range(u + 1, ei):
                    MA = 0
                    MB = 0
                    for a in ci_:
                        MA |= cn[a] & tnodes[int(tm[u])]
                        MB |= cn[a] & tnodes[int(tm[v])]
                    tot["self_unique"] += bin(MA).count("1") * bin(MB).count("1")
            for j in range(i + 1, nl):
                sj, ej, cj_ = levels[j]
                ldist = np.sqrt(((geo[i][0] - geo[j][0]) ** 2).sum(0))
                lsize = geo[i][1] + geo[j][1]
                near = np.zeros((len(ci_), len(cj_)), dtype=bool)
                for x, a in enumerate(ci_):
                    for y, b in enumerate(cj_):
                        near[x, y] = np.any(~((np.abs(ldist - cdist[a, b]) - lsize) > csize[a, b]))
                tot["entries"] += near.size
                tot["near"] += int(near.sum())
                tot["lig_pairs"] += (ei - si) * (ej - sj)
                for u in range(si, ei):
                    tu = tnodes[int(tm[u])]
                    for v in range(sj, ej):
                        tv = tnodes[int(tm[v])]
                        uniq = set()
                        MAall = 0
                        NBall = 0
                        for x, a in enumerate(ci_):
                            A = cn[a] & tu
                            na = bin(A).count("1")
                            hist_listlen[na] += 1
                            MAall |= A
                            for y, b in enumerate(cj_):
                                B = cn[b] & tv
                                nb = bin(B).count("1")
                                if x == 0:
                                    NBall |= B
                                tot["direct_all"] += na * nb
                                if na and nb:
                                    tot["items"] += 1
                                if near[x, y] and na and nb:
                                    tot["items_near"] += 1
                                    tot["direct"] += na * nb
                                    uniq.add((A, B))
                        # unique (m,n) among near entries
                        seen = set()
                        for (A, B) in uniq:
                            ms = [m for m in range(Nm) if (A >> m) & 1]
                            ns = [m for m in range(Nm) if (B >> m) & 1]
                            for m in ms:
                                for nn in ns:
                                    seen.add((m, nn))
                        tot["unique"] += len(seen)
                        nM, nN = bin(MAall).count("1"), bin(NBall).count("1")
                        tot["unique_all"] += nM * nN
                        SB = sum(bin(cn[b] & tv).count("1") for b in cj_)
                        SA = sum(bin(cn[a] & tu).count("1") for a in ci_)
                        tot["rowsum_adds"] += nM * SB
                        tot["entry_adds"] += SA * len(cj_)
    N = len(lib)
    print(f"{N} ligands, {args.conformers} conformers; per ligand (= per ligand-conformer for term counts):")
    for k, v in tot.items():
        print(f"  {k:12s} {v / N:12.1f}")
    print(f"  duplication near: direct/unique = {tot['direct'] / max(tot['unique'], 1):.2f}; "
          f"direct/unique_all = {tot['direct'] / max(tot['unique_all'], 1):.2f}; near fraction {tot['near'] / max(tot['entries'], 1):.3f}")
    print(f"  self: direct {tot['self_direct'] / N:.1f} unique {tot['self_unique'] / N:.1f}")
    nz = np.nonzero(hist_listlen)[0]
    print("  node-list length histogram (|A_a ∩ T(u)|):", {int(k): int(hist_listlen[k]) for k in nz})
    lane_mapping(lib, fm, cn, tnodes, K)


def lane_mapping(lib, fm, cn, tnodes, K, slots=8, chunk=32):
    """tables_kernel_v2 gives the items (entry, u, v) of a cluster pair to its 8 lane groups in order, 32 entries at a
    time; a wave step costs, row by row, the widest padded column list among the groups still having that row."""
    pc = lambda x: bin(x).count("1")
    ideal = 0
    cost = {1: 0, 2: 0, 3: 0}
    for li in range(len(lib)):
        r = lib.unpack(li)
        tm, ends = r["typemask"], r["cluster_end"]
        levels, start = [], 0
        for ci in range(r["n_clusters"]):
            end, lmask = int(ends[ci]), 0
            for u in range(start, end):
                lmask |= int(tm[u])
            cand = [a for a in range(K) if int(fm.cluster_typemask[a]) & lmask]
            if cand and len(levels) < 20:
                levels.append((start, end, cand))
            start = end
        for i in range(len(levels)):
            si, ei, ci_ = levels[i]
            for j in range(i + 1, len(levels)):
                sj, ej, cj_ = levels[j]
                entries = [(a, b) for a in ci_ for b in cj_]
                for e0 in range(0, len(entries), chunk):
                    items = [(pc(cn[a] & tnodes[int(tm[u])]), pc(cn[b] & tnodes[int(tm[v])]))
                             for (a, b) in entries[e0:e0 + chunk] for u in range(si, ei) for v in range(sj, ej)]
                    ideal += sum(na * nb for na, nb in items)
                    for w0 in range(0, len(items), slots):
                        grp = [x for x in items[w0:w0 + slots] if x[0] * x[1]]
                        if not grp:
                            continue
                        for W in cost:
                            for row in range(max(x[0] for x in grp)):
                                cost[W] += max(W * ((x[1] + W - 1) // W) for x in grp if x[0] > row)
    n = len(lib)
    print(f"  lane mapping of tables_kernel_v2 (pair tables, prefilter ignored): {ideal / n / slots:.0f} wave-terms per ligand if every slot carried a term;")
    for W in sorted(cost):
        print(f"    column batches of {W}: {cost[W] / n:.0f} term slots per ligand, {ideal / slots / cost[W]:.1%} filled")


if __name__ == "__main__":
    main()
