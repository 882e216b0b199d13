#!/usr/bin/env python3
"""The reference's own throughput on this path, measured where the reference can run: the BUILD container.

    python tools/reference_rate.py            # writes profiles/reference_numpy_rate.json

`GraphMatcher(model, ligand, weights).run()` (src/pmnet/scoring/graph_match.py:94-101, NumPy kernels of match_utils.py - Numba is not
installable here) imported from /root/reference/src, one process, on the ligands of two golden sets (the 6OIM-like model, 8 conformers;
the molecules come from tests/golden/*_mols.npz and go through the reference's real `LigandGraph`). Timed: `run()` alone (graph in, score
out), as SURVEY.md 8d-ii asks; the `LigandGraph` construction is timed beside it. The GPU box has no /root/reference: bench.py quotes this
file as a static number (`cpu_baseline.reference_numpy_path`), it never runs the reference."""
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests" / "golden"))

import make_golden as mg  # noqa: E402  (stubs openbabel, imports the reference)
from pharmaconet_amd.library import LigandFeatures  # noqa: E402


def load_mols(name):
    d = np.load(REPO / "tests" / "golden" / f"{name}_mols.npz")
    topo = json.loads(str(d["topology"]))
    shapes, flat = d["shapes"], d["positions"]
    out, o = [], 0
    for t, shp in zip(topo, shapes):
        n = int(np.prod(shp))
        pos = flat[o : o + n].reshape(tuple(int(x) for x in shp))
        o += n
        feats = [(f[0], f[1] if isinstance(f[1], int) else tuple(f[1]), f[2] if isinstance(f[2], int) else tuple(f[2])) for f in t["features"]]
        out.append(LigandFeatures(t["z"], t["nbrs"], feats, pos))
    return out


def main():
    import os
    import platform

    model = mg.RefModel.load(str(REPO / "tests" / "golden" / "model_6oim_like.pm"))
    rows = {}
    for name in ("set_6oim_c8", "set_6oim_c8_weights"):
        golden = np.load(REPO / "tests" / "golden" / f"{name}.npz")
        weights = json.loads(str(golden["weights"]))
        mols = load_mols(name)
        t0 = time.perf_counter()
        ligs = [mg.FakeLigand(m) for m in mols]
        t_graph = time.perf_counter() - t0
        t0 = time.perf_counter()
        scores = [mg.GraphMatcher(model, lig, weights).run() for lig in ligs]
        t_run = time.perf_counter() - t0
        assert np.array_equal(np.asarray(scores, dtype=np.float64), golden["score"]), "the reference no longer reproduces the golden scores"
        n_conf = sum(lig.num_conformers for lig in ligs)
        rows[name] = {"ligands": len(ligs), "ligand_conformers": n_conf, "run_s": t_run, "ligand_graph_s": t_graph,
                      "ligand_conformers_per_s_run_only": n_conf / t_run, "ligand_conformers_per_s_with_graph_build": n_conf / (t_run + t_graph),
                      "mean_tree_nodes": float(golden["n_tree"].mean())}
        print(name, rows[name])
    tot_conf = sum(r["ligand_conformers"] for r in rows.values())
    tot_run = sum(r["run_s"] for r in rows.values())
    cpu = ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    out = {
        "what": "GraphMatcher.run() of /root/reference/src/pmnet (NumPy kernels, match_utils.py), ONE process, golden ligands of the 6OIM-like model x 8 conformers; "
                "scores reproduced bit for bit; measured in the build container (the reference cannot travel to the GPU box)",
        "value": tot_conf / tot_run,
        "unit": "ligand-conformers/s",
        "cores": 1,
        "kind": "reference",
        "where": f"build container, {cpu}, {os.cpu_count()} hardware threads visible, python {platform.python_version()}, numpy {np.__version__}",
        "sets": rows,
        "note": "the golden ligands are half active-like (denser trees than the bench library's 10 %): a lower bound of the reference's rate on the bench library; "
                "screening.py --cpus N is N such processes (multiprocessing.Pool, screening.py:66-68)",
    }
    (REPO / "profiles" / "reference_numpy_rate.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: out[k] for k in ("value", "unit", "where")}))


if __name__ == "__main__":
    main()
