"""Frames a branch-and-bound walker enters under the engine's per-candidate bound and under a path-aware bound (bound_study.c)."""
import sys, ctypes, numpy as np, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN
from oracle import oracle
from pharmaconet_amd import PharmacophoreModel
from pharmaconet_amd.constants import weights_vector, TYPE_ID
from tools.synthetic import synthetic_library, BASE_SEED
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libbound_study.so"
os.system(f"gcc -O2 -fopenmp -shared -fPIC {here}/bound_study.c -o {so} -lm")
# run_study.py [model.pm [ligands [conformers]]]: 8 conformers = the bench library (drawn on the 6OIM-like model's nodes);
# 64 conformers = the library of tools/stress_shape.py (drawn on the model's own nodes): run_study.py model_stress64.pm 24 64
name = sys.argv[1] if len(sys.argv) > 1 else "model_6oim_like.pm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nconf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
model = PharmacophoreModel.load(GOLDEN / name)
st = (PharmacophoreModel.load(GOLDEN / "model_6oim_like.pm") if nconf == 8 else model).__getstate__()
centers = np.array([x["center"] for x in st["nodes"]], dtype=np.float64); types = np.array([TYPE_ID[x["type"]] for x in st["nodes"]])
survey = len(sys.argv) > 4 and sys.argv[4] == "survey"  # run_study.py model n 8 survey: SURVEY 8d-2's library (tools/survey_library.py)
if survey:
    from pharmaconet_amd import PackedLibrary
    from tools.survey_library import survey_library
    o_, d_, _ = survey_library(centers, types, n, nconf, "cpu")
    lib = PackedLibrary(o_.numpy().astype(np.uint64), d_.numpy())
elif nconf == 8:
    lib = synthetic_library(n, first=0, num_conformers=8, model_nodes=(centers, types), active_fraction=0.1, seed=BASE_SEED, max_nodes=32)
else:
    lib = synthetic_library(n, num_conformers=nconf, model_nodes=(centers, types), active_fraction=0.2, seed=6464, max_nodes=32, conformer_noise=0.0)
L = ctypes.CDLL(so)
flat = model.flat
keep = dict(node_type=np.ascontiguousarray(flat.node_type, np.uint8), edge_mean=np.ascontiguousarray(flat.edge_mean, np.float32), edge_std=np.ascontiguousarray(flat.edge_std, np.float32), cluster_nodes=np.ascontiguousarray(flat.cluster_nodes, np.uint64), cluster_typemask=np.ascontiguousarray(flat.cluster_typemask, np.uint8), cluster_center=np.ascontiguousarray(flat.cluster_center, np.float64), cluster_size=np.ascontiguousarray(flat.cluster_size, np.float64))
M = oracle.OracleModel(flat.num_nodes, flat.num_clusters, *(keep[k].ctypes.data for k in ("node_type", "edge_mean", "edge_std", "cluster_nodes", "cluster_typemask", "cluster_center", "cluster_size")))
off = np.ascontiguousarray(lib.offsets, np.uint64); dat = np.ascontiguousarray(lib.data, np.uint8)
w = np.array(weights_vector(None), np.float32)
out = np.zeros((n, 16), np.int64)
L.bound_study.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
t0 = time.time(); L.bound_study(ctypes.byref(M), off.ctypes.data, dat.ctypes.data, 0, n, w.ctypes.data, out.ctypes.data)
print(name, "ligands", n, "time", round(time.time() - t0, 1), "score mismatches", int((out[:, 15] != 0).sum()))
print("reference tree nodes per ligand", out[:, 0].mean())
for mode, label in enumerate(("A engine bound", "B path bound everywhere", "B only under >= 5 matches", "B where >= 2 levels lie below the child",
                              "level bound R under >= 5 matches, index order", "W everywhere, index order", "W under >= 5 matches, index order")):
    print(f"{label:45s} frames {out[:, 1 + 2 * mode].mean():9.1f}  path-bound evaluations {out[:, 2 + 2 * mode].mean():9.1f}")
