#!/usr/bin/env python3
"""Randomised parity soak (GPU box): models x conformer counts x type weights drawn at random, SURVEY 8d-2-style ligands drawn on the model's own nodes,
the GPU's scores against the CPU oracle's at the parity tests' tolerance. `python tests/fuzz_parity.py [rounds] [ligands]` (it lives under tests/: only tests, smoke() and bench.py's checker legs may call the oracle)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from oracle import oracle as orc  # noqa: E402
from pharmaconet_amd import PackedLibrary, PharmacophoreModel  # noqa: E402
from pharmaconet_amd.constants import TYPE_ID, weights_vector  # noqa: E402
from pharmaconet_amd.engine import DeviceLibrary  # noqa: E402
from tools.survey_library import survey_library  # noqa: E402

orc.build()
G = REPO / "tests" / "golden"
models = [G / "model_6oim_like.pm", G / "model_clustered21.pm", G / "model_stress64.pm", G / "model_large110.pm"] + sorted((G / "pockets16").glob("model_*.pm"))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_lig = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(606)
names = ("Hydrophobic", "Aromatic", "Cation", "Anion", "HBond_donor", "HBond_acceptor", "Halogen")
worst = 0.0
for r in range(rounds):
    mp = models[int(rng.integers(len(models)))]
    model = PharmacophoreModel.load(mp)
    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    C = int(rng.choice([1, 2, 3, 5, 8, 8, 8, 13, 16, 33, 64]))
    weights = None
    if rng.random() < 0.4:
        weights = {k: float(10 ** rng.uniform(-1, 2)) for k in names if rng.random() < 0.7}
    n = n_lig if C <= 16 else max(100, n_lig // 8)
    off, data, _ = survey_library(centers, types, n, C, "cuda", seed=int(rng.integers(1 << 30)), active_fraction=float(rng.choice([0.0, 0.1, 0.5, 1.0])))
    dlib = DeviceLibrary.from_device_buffers(off, data)
    t0 = time.time()
    res = model.screen(dlib, weights=weights)
    got = res.scores.cpu().numpy().astype(np.float64)
    status = res.status.cpu().numpy()
    lib = PackedLibrary(off.cpu().numpy().astype(np.uint64), data.cpu().numpy())
    # (the oracle walks whole trees - a few ligands of a heavy draw take minutes: it runs in a child process with a time limit, and a round that exceeds it is skipped)
    import multiprocessing as mp_

    def _work(q):
        q.put(orc.oracle_score(model.flat, lib, weights_vector(weights), num_threads=16))

    q = mp_.Queue()
    child = mp_.Process(target=_work, args=(q,))
    child.start()
    try:
        ref = q.get(timeout=float(sys.argv[3]) if len(sys.argv) > 3 else 25.0)
    except Exception:
        child.kill()
        child.join()
        print(json.dumps({"round": r, "model": mp.name, "C": C, "skipped": "oracle over its time limit"}), flush=True)
        dlib.close()
        continue
    child.join()
    ok = status == 0
    nz = ok & (ref > 0)
    err = np.abs(got[nz] - ref[nz]) / ref[nz]
    zero_bad = int(((got[ok & (ref == 0)]) != 0).sum())
    m = float(err.max()) if nz.any() else 0.0
    worst = max(worst, m)
    print(json.dumps({"round": r, "model": mp.name, "C": C, "weights": weights, "ligands": n, "scored": int(ok.sum()), "nonzero": int(nz.sum()), "max_rel_err": m, "zero_mismatch": zero_bad,
                      "s": round(time.time() - t0, 1)}), flush=True)
    assert m < 2e-6 + 6e-8 and zero_bad == 0, "PARITY FAILURE"
    dlib.close()
print("fuzz ok: worst", worst)
