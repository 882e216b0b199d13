#!/usr/bin/env python3
"""Mint the multi-pocket fixture (BASELINE.json configs[3]: a batch of 16 distinct pockets x one shared library) by
running the REFERENCE itself. Build container only (needs /root/reference):

    python tests/golden/make_golden_pockets.py

16 synthetic pharmacophore models (25-55 hotspots, different seeds) are built through the real
`PharmacophoreModel.create` and saved with the real `save()`; one shared set of feature molecules goes through the real
`LigandGraph` and is scored with the real `GraphMatcher.run()` against every model. Written: `pockets16/model_XX.pm`,
`pockets16.pmxlib` (the packed library extracted from the real `LigandGraph`s) and `pockets16.npz` (scores [16][n]).
Only data is written.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as mg  # noqa: E402  (stubs openbabel, imports the reference)

N_POCKETS, N_LIGANDS, N_CONF = 16, 64, 8


def main():
    out = HERE / "pockets16"
    out.mkdir(exist_ok=True)
    models = []
    for k in range(N_POCKETS):
        rng = np.random.default_rng(mg.SEED + 7000 + k)
        n_hot = int(rng.integers(25, 56))
        model = mg.model_random(mg.SEED + 7100 + k, n_hot, extent=6.0 + 0.2 * k, name=f"POCKET {k:02d}", min_sep=1.0)
        model.save(str(out / f"model_{k:02d}.pm"))
        models.append(model)
        print(f"pocket {k:02d}: {len(model.nodes)} nodes, {len(model.node_clusters)} clusters")
    # shared library: half of the ligands are drawn on pocket 0's nodes, the rest on the others' (so that every pocket has actives)
    mols = []
    for i in range(N_LIGANDS):
        rng = mg.ligand_rng(mg.SEED + 7200, i)
        mn = mg.model_nodes_of(models[i % N_POCKETS])
        while True:
            m = mg.random_molecule(rng, N_CONF, model_nodes=mn, active_like=rng.random() < 0.6)
            lig = mg.FakeLigand(m)
            if len(lig.graph.nodes) <= 32 and len(lig.graph.node_clusters) <= 64:
                break
        mols.append(m)
    ligs = [mg.FakeLigand(m) for m in mols]
    lib = mg.PackedLibrary.from_records([mg.pack_clustered_ligand(mg.extract(l.graph)) for l in ligs])
    lib.save(out.parent / "pockets16.pmxlib")
    scores = np.zeros((N_POCKETS, N_LIGANDS))
    t0 = time.time()
    for k, model in enumerate(models):
        for i, lig in enumerate(ligs):
            scores[k, i] = mg.reference_run(model, lig, None)["score"]
        print(f"pocket {k:02d}: mean score {scores[k].mean():.2f}, nonzero {np.count_nonzero(scores[k])}, {time.time() - t0:.0f}s")
    np.savez_compressed(out.parent / "pockets16.npz", score=scores, n_nodes=np.array([len(m.nodes) for m in models]),
                        n_clusters=np.array([len(m.node_clusters) for m in models]))


if __name__ == "__main__":
    main()
