"""The synthetic library of SURVEY.md section 8d-2, generated where it is used (torch, any device), as packed records.

    n_nodes ~ clip(round(N(20, 6)), 4, 32); a cluster structure that mimics `LigandGraph` output - aromatic rings with 0-4 dependents,
    charged groups with their polar neighbours, 1-4-node HBond and hydrophobic groups, halogen singletons - with a node type mix of about
    Hydrophobic 45 %, HBA 20 %, HBD 10 %, Aromatic 12 %, Halogen 5 %, Cation 4 %, Anion 4 %; n_conf conformers = base coordinates +
    N(0, 0.5 A) per node and conformer; 10 % "active-like" ligands whose base coordinates are type-compatible model node centres +
    N(0, 0.7 A) under a random rigid motion, 90 % random walks (3-4 A steps between clusters, 1.3-2.6 A inside one).

Every ligand draws from its OWN counter-based stream keyed by (seed, global ligand index) - 20250523 and the ligand's number - so any
shard of the library can be produced independently (rank r of a sharded run makes ligands [r N, (r + 1) N)) and no two ligands share
anything, unlike `tools/synthetic.expand_library_on_device` (4 096 topologies x jittered copies). The records are written directly in the
packed format of `pharmaconet_amd/library.py` (clusters in `priority_fn` order, `graph_match.py:43-60`; the high-priority node first in its
cluster, `ligand.py:387-395`); there are no molecules behind them, so the packer leg of the bench does not apply to this library.

Nothing here mirrors reference code: the reference ships no generator."""

from __future__ import annotations

import math

import numpy as np

SEED = 20250523
MAXN = 32   # nodes per ligand at most (BASELINE configs[1]: <= 32 pharmacophore points)
MAXC = 32   # clusters per ligand at most (every node its own cluster)

# type ids (include/pmx.h): Hydrophobic 0, Aromatic 1, Cation 2, Anion 3, HBond_donor 4, HBond_acceptor 5, Halogen 6
HYD, ARO, CAT, ANI, HBD, HBA, HAL = (1 << t for t in range(7))
# cluster kinds in priority_fn's (group, subtype) order: Aromatic, Cation, Anion | HBond, Halogen, Hydrophobic
K_RING, K_CAT, K_ANI, K_HB, K_HAL, K_HYD = range(6)
KIND_GROUP = (0, 0, 0, 1, 1, 1)
KIND_SUB = (0, 1, 2, 0, 1, 2)
# cluster kind probabilities and sizes chosen so that the NODE type mix comes out at the survey's (checked by tests/test_survey_library.py)
KIND_P = (0.248, 0.083, 0.083, 0.191, 0.103, 0.292)

_M64 = (1 << 64) - 1


def _i64(x: int) -> int:
    """A 64-bit pattern as the int64 torch holds."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_C1, _C2, _GOLD = _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB), _i64(0x9E3779B97F4A7C15)


def _mix(z):
    """splitmix64's finalizer on int64 tensors (multiplications wrap, shifts made logical)."""
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _C1
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _C2
    return z ^ ((z >> 31) & ((1 << 33) - 1))


class _Stream:
    """Counter-based uniform / normal draws per ligand: value = f(key[ligand], counter). Counters are handed out in a fixed order, so a
    ligand's draws do not depend on which other ligands are generated with it."""

    def __init__(self, torch, seed: int, ligand_index):
        self.torch = torch
        self.key = _mix(ligand_index * _GOLD + _i64(seed * 0xD1342543DE82EF95 + 0x632BE59BD9B4E019))
        self.ctr = 0

    def bits(self, *shape):
        t = self.torch
        m = int(np.prod(shape)) if shape else 1
        c = t.arange(self.ctr + 1, self.ctr + 1 + m, device=self.key.device, dtype=t.int64)
        self.ctr += m
        z = _mix(self.key[:, None] + c[None, :] * _GOLD)
        return z.reshape(self.key.shape[0], *shape)

    def uniform(self, *shape):
        """float64 in (0, 1): the top 53 bits, never 0."""
        return (((self.bits(*shape) >> 11) & ((1 << 53) - 1)).to(self.torch.float64) + 0.5) * (1.0 / (1 << 53))

    def normal(self, *shape):
        t = self.torch
        u1, u2 = self.uniform(*shape), self.uniform(*shape)
        return t.sqrt(-2.0 * t.log(u1)) * t.cos((2.0 * math.pi) * u2)


def _choice(torch, u, probs):
    """Index drawn from `probs` by the uniform `u` (a tensor)."""
    edges = torch.tensor(np.cumsum(probs)[:-1], dtype=u.dtype, device=u.device)
    return torch.bucketize(u, edges)


def _generate_chunk(torch, model_centers, model_types, first: int, count: int, n_conf: int, device, seed: int, active_fraction: float):
    """Ligands [first, first + count): per-ligand byte counts and the tensors a record is assembled from."""
    dev = torch.device(device)
    lig = torch.arange(first, first + count, device=dev, dtype=torch.int64)
    rs = _Stream(torch, seed, lig)
    N = count
    # ---- size and kind
    n = torch.clamp(torch.round(20.0 + 6.0 * rs.normal(1)[:, 0]), 4, MAXN).to(torch.int64)
    active = rs.uniform(1)[:, 0] < active_fraction
    # ---- clusters in generation order: kind, size; cut where the node budget n ends (the last cluster is truncated)
    kind = _choice(torch, rs.uniform(MAXC), KIND_P)                                   # [N, MAXC]
    us = rs.uniform(MAXC)
    size = torch.ones_like(kind)
    size = torch.where(kind == K_RING, 1 + torch.floor(us * 5).to(torch.int64), size)                     # ring + 0..4 dependents
    size = torch.where((kind == K_CAT) | (kind == K_ANI), 1 + torch.floor(us * 3).to(torch.int64), size)  # charged group + 0..2
    size = torch.where(kind == K_HB, 1 + _choice(torch, us, (0.6, 0.25, 0.1, 0.05)), size)
    size = torch.where(kind == K_HYD, 1 + _choice(torch, us, (0.4, 0.3, 0.2, 0.1)), size)
    start = torch.cumsum(size, 1) - size
    size = torch.clamp(torch.minimum(size, n[:, None] - start), min=0)                # clusters past the budget: 0 nodes
    k = (size > 0).sum(1)                                                             # clusters of the ligand
    # ---- nodes in generation order: cluster, rank inside it, type mask
    node = torch.arange(MAXN, device=dev, dtype=torch.int64)[None, :].expand(N, MAXN)
    end = start + size
    ncl = (node[:, :, None] >= end[:, None, :]).sum(2).clamp(max=MAXC - 1)            # cluster of node i (generation order)
    nrank = node - torch.gather(start, 1, ncl)
    nkind = torch.gather(kind, 1, ncl)
    ut = rs.uniform(MAXN)
    head = nrank == 0
    tm = torch.full((N, MAXN), HYD, dtype=torch.int64, device=dev)
    dep_ring = torch.where(ut < 0.7, HYD, HBA)                                        # ring carbons / a ring nitrogen
    hb = torch.where(ut < 0.55, HBA, torch.where(ut < 0.90, HBD, HBD | HBA))          # carbonyl / amine / hydroxyl
    tm = torch.where(nkind == K_RING, torch.where(head, ARO, dep_ring), tm)
    tm = torch.where(nkind == K_CAT, torch.where(head, torch.where(ut < 0.6, CAT | HBA, CAT), HBD), tm)  # (a tertiary amine is Cation + acceptor)
    tm = torch.where(nkind == K_ANI, torch.where(head, ANI, HBA), tm)
    tm = torch.where(nkind == K_HB, hb, tm)
    tm = torch.where(nkind == K_HAL, HAL, tm)
    # ---- base coordinates
    def unit(shape_tail):
        v = rs.normal(*shape_tail, 3)
        return v / torch.sqrt((v * v).sum(-1, keepdim=True)).clamp(min=1e-12)

    # random walk: heads step 3-4 A from the head before, members sit 1.3-2.6 A from their head
    hstep = unit((MAXC,)) * (3.0 + rs.uniform(MAXC))[..., None]                       # [N, MAXC, 3]
    drift = unit(())                                                                   # a preferred direction keeps the walk from curling up
    hstep = hstep + 1.2 * drift[:, None, :]
    hpos = torch.cumsum(hstep, 1)
    mstep = unit((MAXN,)) * (1.3 + 1.3 * rs.uniform(MAXN))[..., None]
    walk = torch.gather(hpos, 1, ncl[:, :, None].expand(N, MAXN, 3)) + torch.where(head[:, :, None], torch.zeros_like(mstep), mstep)
    # active-like: a type-compatible model node's centre + N(0, 0.7), then one random rigid motion of the whole ligand
    mc = torch.as_tensor(np.asarray(model_centers, dtype=np.float64), device=dev)
    mt = torch.as_tensor(np.asarray(model_types, dtype=np.int64), device=dev)
    compat = ((tm[:, :, None] >> mt[None, None, :]) & 1).to(torch.float64)           # [N, MAXN, Nm]
    compat = torch.where(compat.sum(2, keepdim=True) > 0, compat, torch.ones_like(compat))
    cdf = torch.cumsum(compat, 2)
    pick = (rs.uniform(MAXN)[:, :, None] * cdf[:, :, -1:] >= cdf).sum(2).clamp(max=mc.shape[0] - 1)
    on_model = mc[pick] + 0.7 * rs.normal(MAXN, 3)
    q = rs.normal(4)
    q = q / torch.sqrt((q * q).sum(1, keepdim=True)).clamp(min=1e-12)
    a, b, c, d = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rot = torch.stack([a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
                       2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
                       2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], 1).reshape(N, 3, 3)
    shift = 40.0 * (rs.uniform(3) - 0.5)
    centre = mc.mean(0)
    moved = torch.einsum("nij,nmj->nmi", rot, on_model - centre) + centre + shift[:, None, :]
    base = torch.where(active[:, None, None], moved, walk + centre)                   # [N, MAXN, 3]
    xyz = base[:, :, None, :] + 0.5 * rs.normal(MAXN, n_conf, 3)                      # [N, MAXN, C, 3]
    # ---- priority_fn order of the clusters (graph_match.py:43-60): (group, -size, subtype, key atom = generation index), empty ones last
    grp = torch.tensor(KIND_GROUP, device=dev)[kind]
    sub = torch.tensor(KIND_SUB, device=dev)[kind]
    cidx = torch.arange(MAXC, device=dev, dtype=torch.int64)[None, :]
    key = torch.where(size > 0, ((grp * 64 + (63 - size)) * 8 + sub) * 64 + cidx, torch.full_like(size, 1 << 40))
    order = torch.argsort(key, dim=1)                                                 # sorted position -> generation cluster
    pos = torch.empty_like(order)
    pos.scatter_(1, order, cidx.expand(N, MAXC).contiguous())                         # generation cluster -> sorted position
    ssize = torch.gather(size, 1, order)
    send = torch.cumsum(ssize, 1)                                                     # cluster_end in sorted order
    sstart = send - ssize
    new_index = torch.gather(sstart, 1, torch.gather(pos, 1, ncl)) + nrank            # node (generation order) -> node (record order)
    live = node < n[:, None]
    new_index = torch.where(live, new_index, node)                                    # (dead nodes keep their own slot: never written)
    return dict(n=n, k=k, tm=tm, xyz=xyz.to(torch.float32), new_index=new_index, live=live, send=send, active=active)


def survey_library(model_centers, model_types, n_ligands: int, n_conf: int = 8, device="cpu", seed: int = SEED, first: int = 0,
                   active_fraction: float = 0.1, chunk: int = 65536):
    """Ligands [first, first + n_ligands) of the survey library against a model whose nodes are (`model_centers` [Nm, 3], `model_types` [Nm]
    type ids). Returns `(offsets int64 [n_ligands + 1], data uint8)` torch tensors on `device` in the packed library format, plus a
    dict of statistics (node type mix, mean nodes, clusters, active share)."""
    import torch

    dev = torch.device(device)
    C = int(n_conf)
    sizes, parts = [], []
    stats = {"nodes": 0, "clusters": 0, "active": 0, "type_nodes": np.zeros(7, dtype=np.int64)}
    # two passes per chunk would double the work: a chunk's tensors are kept until its bytes are placed (a chunk is ~0.4 GB at 8 conformers)
    for lo in range(0, n_ligands, chunk):
        cnt = min(chunk, n_ligands - lo)
        g = _generate_chunk(torch, model_centers, model_types, first + lo, cnt, C, dev, seed, active_fraction)
        n, k = g["n"], g["k"]
        head = (8 + n + k + 3) & ~3
        rec = (head + 12 * n * C + 15) & ~15
        off = torch.cumsum(rec, 0) - rec
        total = int(rec.sum())
        data = torch.zeros(total, dtype=torch.uint8, device=dev)
        d16, d32 = data.view(torch.int16), data.view(torch.float32)
        h = off // 2
        d16[h] = n.to(torch.int16)
        d16[h + 1] = torch.full_like(n, C).to(torch.int16)
        d16[h + 2] = k.to(torch.int16)
        live = g["live"]
        # type masks, in record order
        idx = (off[:, None] + 8 + g["new_index"])[live]
        data[idx] = g["tm"][live].to(torch.uint8)
        # cluster ends
        cl = torch.arange(MAXC, device=dev)[None, :] < k[:, None]
        idx = (off[:, None] + 8 + n[:, None] + torch.arange(MAXC, device=dev)[None, :])[cl]
        data[idx] = g["send"][cl].to(torch.uint8)
        # coordinates: xyz[node][axis][conformer]
        base = ((off + head) // 4)[:, None] + g["new_index"] * (3 * C)                # [N, MAXN] float index of a node's block
        ax = torch.arange(3, device=dev)[None, None, None, :] * C
        cf = torch.arange(C, device=dev)[None, None, :, None]
        idx = (base[:, :, None, None] + ax + cf)[live]                                # [live nodes, C, 3]
        d32[idx.reshape(-1)] = g["xyz"][live].reshape(-1)
        sizes.append(rec)
        parts.append(data)
        stats["nodes"] += int(n.sum())
        stats["clusters"] += int(k.sum())
        stats["active"] += int(g["active"].sum())
        tml = g["tm"][live]
        for t in range(7):
            stats["type_nodes"][t] += int(((tml >> t) & 1).sum())
        del g
    rec_all = torch.cat(sizes) if sizes else torch.zeros(0, dtype=torch.int64, device=dev)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(rec_all, 0)])
    data = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)
    out = {
        "mean_nodes": stats["nodes"] / max(n_ligands, 1),
        "mean_clusters": stats["clusters"] / max(n_ligands, 1),
        "active_share": stats["active"] / max(n_ligands, 1),
        "type_share_of_nodes": dict(zip(("Hydrophobic", "Aromatic", "Cation", "Anion", "HBond_donor", "HBond_acceptor", "Halogen"),
                                        (stats["type_nodes"] / max(stats["nodes"], 1)).round(4).tolist())),
    }
    return offsets, data, out
