#!/usr/bin/env python3
"""Mint the golden fixtures of tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

What it does (procedure of SURVEY.md Appendix B):
  * imports `pmnet` from /root/reference/src with `openbabel` stubbed by a MagicMock - numba is
    absent, so `graph_match.py:12-15` binds the NumPy kernels of `match_utils.py` (the canonical
    variant for parity);
  * builds synthetic pharmacophore models through the real `PharmacophoreModel.create`
    (`pharmacophore_model.py:108-149` -> `utils/density_map.py`) and saves them with the real
    `save()` as `.pm` / `.json`;
  * draws synthetic feature molecules (`tools.synthetic`), wraps each in a fake ligand
    object, builds the real `LigandGraph` (`scoring/ligand.py:110-259`) and scores it with the real
    `GraphMatcher(...).run()` (`scoring/graph_match.py:94-101`);
  * writes, per ligand set: the packed library extracted from the real `LigandGraph`
    (`*.pmxlib`), the feature molecules (`*_mols.json`, input of the packer tests) and an `.npz`
    with the reference's scores, level / tree / leaf counts and pair-table checksums.

Only data is written: no reference source text or bytecode ends up in the fixtures.
"""

from __future__ import annotations

import json
import math
import sys
import time
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/src")

_ob = MagicMock()
sys.modules["openbabel"] = _ob
sys.modules["openbabel.pybel"] = _ob.pybel
_ob.pybel.ob.OBAtomAtomIter = lambda atom: iter(atom.nbrs)

from pmnet.pharmacophore_model import PharmacophoreModel as RefModel  # noqa: E402
from pmnet.scoring import graph_match as ref_gm  # noqa: E402
from pmnet.scoring.graph_match import GraphMatcher  # noqa: E402
from pmnet.scoring.ligand import LigandGraph  # noqa: E402
from pmnet.scoring.ligand_utils import PharmacophoreNode  # noqa: E402

from pharmaconet_amd.constants import TYPE_ID  # noqa: E402
from pharmaconet_amd.library import ClusteredLigand, LigandFeatures, PackedLibrary, pack_clustered_ligand  # noqa: E402
from tools.synthetic import ligand_rng, random_molecule  # noqa: E402

assert ref_gm.scoring_matching_pair.__module__ == "pmnet.scoring.match_utils", "expected the NumPy kernels"

SEED = 20250523
GRID, RES = 64, 0.5


# ----------------------------------------------------------------------------- models
def blob(center_xyz, box_center, radius_vox):
    """Gaussian blob on the 64^3 / 0.5 A grid, thresholded at 0.5."""
    origin = np.asarray(box_center) - RES * (GRID - 1) / 2
    v0 = (np.asarray(center_xyz) - origin) / RES
    s = radius_vox / math.sqrt(2 * math.log(2))
    ax = np.arange(GRID, dtype=np.float64)
    gx = np.exp(-((ax - v0[0]) ** 2) / (2 * s * s))
    gy = np.exp(-((ax - v0[1]) ** 2) / (2 * s * s))
    gz = np.exp(-((ax - v0[2]) ** 2) / (2 * s * s))
    m = gx[:, None, None] * gy[None, :, None] * gz[None, None, :]
    m[m <= 0.5] = 0.0
    return m.astype(np.float32)


def hotspot(nci_type, pos, box_center, rng):
    pos = np.asarray(pos, dtype=np.float32)
    return dict(
        nci_type=nci_type,
        hotspot_position=pos,
        hotspot_score=float(rng.uniform(0.5, 1.0)),
        point_map=blob(pos + rng.normal(scale=0.3, size=3), box_center, rng.uniform(2.0, 3.2)),
    )


def read_6oim_ligand():
    atoms, bonds = [], {}
    for line in open("/root/reference/examples/6OIM_D_MOV.pdb"):
        if line.startswith("HETATM"):
            atoms.append((line[76:78].strip(), float(line[30:38]), float(line[38:46]), float(line[46:54])))
        elif line.startswith("CONECT"):
            ids = [int(x) for x in line.split()[1:]]
            bonds.setdefault(ids[0] - 1, set()).update(i - 1 for i in ids[1:])
    return atoms, bonds


def six_rings(bonds, n):
    rings = set()

    def walk(path):
        if len(path) == 6:
            if path[0] in bonds.get(path[-1], ()):
                rings.add(tuple(sorted(path)))
            return
        for nb in bonds.get(path[-1], ()):
            if nb not in path:
                walk(path + [nb])

    for a in range(n):
        walk([a])
    return sorted(rings)


def model_6oim_like():
    rng = np.random.default_rng(SEED + 1)
    atoms, bonds = read_6oim_ligand()
    xyz = np.array([a[1:] for a in atoms])
    box = tuple(xyz.mean(axis=0).tolist())
    infos = []
    for (el, *_), p in zip(atoms, xyz):
        if el == "C":
            if rng.random() < 0.55:
                infos.append(hotspot("Hydrophobic", p, box, rng))
        elif el == "N":
            infos.append(hotspot("HBond_pdon" if rng.random() < 0.5 else "HBond_ldon", p, box, rng))
        elif el == "O":
            infos.append(hotspot("HBond_pdon", p, box, rng))
        elif el == "F":
            infos.append(hotspot("XBond", p, box, rng))
    for ring in six_rings(bonds, len(atoms))[:3]:
        centroid = xyz[list(ring)].mean(axis=0)
        for nci in ("PiStacking_P", "PiStacking_T", "PiCation_pring"):
            if rng.random() < 0.7:
                infos.append(hotspot(nci, centroid, box, rng))
    infos.append(hotspot("SaltBridge_pneg", xyz[14], box, rng))
    infos.append(hotspot("SaltBridge_lneg", xyz[5], box, rng))
    return RefModel.create("SYNTHETIC 6OIM-LIKE", box, infos)


NCI_MIX = (
    ("Hydrophobic", 0.40),
    ("PiStacking_P", 0.06),
    ("PiStacking_T", 0.05),
    ("PiCation_lring", 0.03),
    ("PiCation_pring", 0.04),
    ("HBond_ldon", 0.12),
    ("HBond_pdon", 0.14),
    ("SaltBridge_lneg", 0.05),
    ("SaltBridge_pneg", 0.05),
    ("XBond", 0.06),
)


def model_random(seed, n_hotspots, extent, name, min_sep=0.0, exclude=()):
    rng = np.random.default_rng(seed)
    names = [n for n, _ in NCI_MIX if n not in exclude]
    probs = np.array([p for n, p in NCI_MIX if n not in exclude])
    probs /= probs.sum()
    box = (10.0, -5.0, 3.0)
    pts = []
    while len(pts) < n_hotspots:
        p = np.array(box) + rng.uniform(-extent, extent, size=3)
        if all(np.linalg.norm(p - q) >= min_sep for q in pts):
            pts.append(p)
    infos = [hotspot(names[int(rng.choice(len(names), p=probs))], p, box, rng) for p in pts]
    return RefModel.create(name, box, infos)


# ---------------------------------------------------------------------------- ligands
class FakeAtom:
    def __init__(self, idx, z):
        self.idx, self.z, self.nbrs = idx, z, []

    def GetIdx(self):
        return self.idx + 1

    def GetAtomicNum(self):
        return self.z


class FakeLigand:
    """Exposes what `LigandGraph.__init__` and `GraphMatcher.__init__` read (`ligand.py:120-132`, `graph_match.py:71-74`)."""

    def __init__(self, feats: LigandFeatures):
        self.obatoms = [FakeAtom(i, z) for i, z in enumerate(feats.atomic_nums)]
        for atom, nbrs in zip(self.obatoms, feats.heavy_neighbors):
            atom.nbrs = [self.obatoms[j] for j in nbrs]
        self.num_atoms = len(self.obatoms)
        self.num_rotatable_bonds = 0
        self.atom_positions = np.asarray(feats.atom_positions, dtype=np.float32)
        self.num_conformers = self.atom_positions.shape[1]
        self.pharmacophore_list = [(t, PharmacophoreNode(a, c)) for t, a, c in feats.features]
        self.graph = LigandGraph(self)


def extract(graph: LigandGraph) -> ClusteredLigand:
    n = len(graph.nodes)
    typemask = np.zeros(n, dtype=np.uint8)
    positions = np.zeros((n, graph.num_conformers, 3), dtype=np.float32)
    for node in graph.nodes:
        for t in node.types:
            typemask[node.index] |= 1 << TYPE_ID[t]
        positions[node.index] = node.positions
    clusters = [[node.index for node in cluster.nodes] for cluster in graph.node_clusters]
    ctypes = [cluster.type for cluster in graph.node_clusters]
    keys = [min(cluster.nodes[0].atom_indices) for cluster in graph.node_clusters]
    return ClusteredLigand(typemask, positions, clusters, ctypes, keys)


def reference_run(model, lig: FakeLigand, weights):
    """`GraphMatcher.run()` (`graph_match.py:94-101`) with the intermediate objects kept for the checksums."""
    gm = GraphMatcher(model, lig, weights)
    out = dict(score=0.0, n_levels=0, n_tree=0, n_leaf=0, s_sum=0.0, p_sum=0.0, p_invalid=0, p_entries=0)
    if len(gm.ligand_graph.node_clusters) == 0:
        return out
    gm.setup()
    out["n_levels"] = len(gm.ligand_cluster_list)
    if len(gm.ligand_cluster_list) == 0:
        return out
    for (lc1, lc2), table in gm.matching_pair_scores_dict.items():
        for values in table.values():
            if lc1 is lc2:
                out["s_sum"] += float(sum(values))
            else:
                for v in values:
                    out["p_entries"] += 1
                    if v < 0:
                        out["p_invalid"] += 1
                    else:
                        out["p_sum"] += float(v)
    root = gm.run_tree()
    out["score"] = gm._run_average(root)

    def count(node):
        return 1 + sum(count(ch) for ch in node.children)

    out["n_tree"] = count(root) - 1
    out["n_leaf"] = sum(1 for _ in root.iteration())
    return out


def feats_to_json(f: LigandFeatures):
    return dict(
        z=[int(x) for x in f.atomic_nums],
        nbrs=[[int(j) for j in row] for row in f.heavy_neighbors],
        features=[
            [t, a if isinstance(a, int) else list(a), c if isinstance(c, int) else list(c)] for t, a, c in f.features
        ],
    )


def model_nodes_of(model):
    centers = np.array([n.center for n in model.nodes], dtype=np.float64)
    types = np.array([TYPE_ID[n.type] for n in model.nodes])
    return centers, types


def make_set(name, model, model_name, count, num_conf, seed, weights=None, active_fraction=0.5, extra=(), big=0):
    t0 = time.time()
    mn = model_nodes_of(model)
    mols: list[LigandFeatures] = []
    for i in range(count):
        rng = ligand_rng(seed, i)
        active = rng.random() < active_fraction
        nfrag = None
        if i < big:  # ligands with > 20 clusters exercise the depth cap (graph_match.py:88)
            nfrag = 26
        while True:
            m = random_molecule(rng, num_conf, n_fragments=nfrag, model_nodes=mn, active_like=active)
            lig = FakeLigand(m)
            if len(lig.graph.nodes) <= 64 and len(lig.graph.node_clusters) <= 64:
                break
            nfrag = (nfrag or 10) - 2
        mols.append(m)
    mols.extend(extra)
    records, rows, positions = [], [], []
    for m in mols:
        lig = FakeLigand(m)
        records.append(pack_clustered_ligand(extract(lig.graph)))
        rows.append(reference_run(model, lig, weights))
        positions.append(np.asarray(m.atom_positions, dtype=np.float32))
    lib = PackedLibrary.from_records(records)
    lib.save(HERE / f"{name}.pmxlib")
    np.savez_compressed(
        HERE / f"{name}.npz",
        model=model_name,
        weights=json.dumps(weights),
        score=np.array([r["score"] for r in rows], dtype=np.float64),
        n_levels=np.array([r["n_levels"] for r in rows], dtype=np.int32),
        n_tree=np.array([r["n_tree"] for r in rows], dtype=np.int64),
        n_leaf=np.array([r["n_leaf"] for r in rows], dtype=np.int64),
        s_sum=np.array([r["s_sum"] for r in rows], dtype=np.float64),
        p_sum=np.array([r["p_sum"] for r in rows], dtype=np.float64),
        p_invalid=np.array([r["p_invalid"] for r in rows], dtype=np.int64),
        p_entries=np.array([r["p_entries"] for r in rows], dtype=np.int64),
    )
    np.savez_compressed(
        HERE / f"{name}_mols.npz",
        topology=json.dumps([feats_to_json(m) for m in mols]),
        positions=np.concatenate([p.reshape(-1) for p in positions]) if positions else np.zeros(0, np.float32),
        shapes=np.array([p.shape for p in positions], dtype=np.int64).reshape(-1, 3),
    )
    sc = np.array([r["score"] for r in rows])
    print(
        f"{name}: {len(mols)} ligands x {num_conf} conf, score mean {sc.mean():.3f} max {sc.max():.3f} "
        f"nonzero {np.count_nonzero(sc)}, tree max {max(r['n_tree'] for r in rows)}, "
        f"levels max {max(r['n_levels'] for r in rows)}, {time.time() - t0:.1f}s"
    )


def special_molecules(num_conf):
    """Edge cases: no features at all; one lone feature; halogens only; a bare carboxylate."""
    rng = np.random.default_rng(SEED + 99)

    def mol(z, bonds, feats):
        nbrs = [[] for _ in z]
        for a, b in bonds:
            nbrs[a].append(b)
            nbrs[b].append(a)
        pos = rng.normal(scale=2.0, size=(len(z), 1, 3)) + rng.normal(scale=0.3, size=(len(z), num_conf, 3))
        return LigandFeatures(z, nbrs, feats, pos.astype(np.float32))

    return [
        mol([6, 8], [(0, 1)], []),  # C-O with the O not flagged: zero features -> score 0 (graph_match.py:95-96)
        mol([6, 6], [(0, 1)], [("Hydrophobic", 0, 0), ("Hydrophobic", 1, 1)]),
        mol([6, 17, 9], [(0, 1), (0, 2)], [("Halogen", 1, 1), ("Halogen", 2, 2)]),
        mol(
            [6, 6, 8, 8],
            [(0, 1), (1, 2), (1, 3)],
            [("Anion", (1, 2, 3), (2, 3)), ("HBond_acceptor", 2, 2), ("HBond_acceptor", 3, 3)],
        ),
    ]


def main():
    models = {
        "model_6oim_like": model_6oim_like(),
        "model_clustered21": model_random(SEED + 2, 21, 6.0, "SYNTHETIC CLUSTERED-21", exclude=("XBond",)),
        "model_stress64": model_random(SEED + 3, 64, 9.0, "SYNTHETIC STRESS-64", min_sep=1.2),
    }
    for name, model in models.items():
        model.save(HERE / f"{name}.pm")
        sizes = [len(c.nodes) for c in model.node_clusters]
        print(f"{name}: {len(model.nodes)} nodes, {len(model.edges)} edges, {len(model.node_clusters)} clusters {sizes}")
    models["model_6oim_like"].save(HERE / "model_6oim_like.json")

    m6, m21, m64 = models["model_6oim_like"], models["model_clustered21"], models["model_stress64"]
    make_set("set_6oim_c8", m6, "model_6oim_like", 300, 8, SEED + 10, extra=special_molecules(8), big=6)
    make_set("set_6oim_c1", m6, "model_6oim_like", 60, 1, SEED + 11, extra=special_molecules(1))
    make_set("set_6oim_c64", m6, "model_6oim_like", 24, 64, SEED + 12)
    make_set("set_6oim_c5", m6, "model_6oim_like", 40, 5, SEED + 13)
    make_set(
        "set_6oim_c8_weights",
        m6,
        "model_6oim_like",
        80,
        8,
        SEED + 10,
        weights=dict(HBond_donor=5.0, HBond_acceptor=5.0, Aromatic=8.0),  # README.md:172
    )
    make_set("set_c21_c8", m21, "model_clustered21", 200, 8, SEED + 20, extra=special_molecules(8))
    make_set("set_s64_c8", m64, "model_stress64", 60, 8, SEED + 30, big=2)
    make_set("set_s64_c64", m64, "model_stress64", 12, 64, SEED + 31)


if __name__ == "__main__":
    main()
