"""Packer throughput against the number of host threads (pmx_pack_features on the bench library's molecule topologies)."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import GOLDEN
from pharmaconet_amd import PharmacophoreModel, _ffi
from pharmaconet_amd.constants import TYPE_ID
from tools.synthetic import synthetic_library, BASE_SEED
from pharmaconet_amd.library import flatten_features
model = PharmacophoreModel.load(GOLDEN / "model_6oim_like.pm")
st = model.__getstate__()
centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
mols = []
synthetic_library(4096, first=0, num_conformers=8, model_nodes=(centers, types), active_fraction=0.1, seed=BASE_SEED, max_nodes=32, conformer_noise=0.0, molecules_out=mols)
flat = flatten_features(mols * int(sys.argv[1] if len(sys.argv) > 1 else 64))
lib = _ffi.load_packer()
n = int(flat["atom_off"].shape[0]) - 1
batch = _ffi.FeatureBatch(n, *(flat[k].ctypes.data for k in ("atom_off", "atomic_num", "nbr_off", "nbr", "feat_off", "feat_type", "feat_flags", "feat_atom_off", "feat_atoms", "feat_center_off", "feat_centers", "n_conf", "pos_off", "positions")))
offsets = np.zeros(n + 1, np.uint64); status = np.zeros(n, np.int32); nb = ctypes.c_uint64(0)
lib.pmx_pack_features(ctypes.byref(batch), 1, offsets.ctypes.data, None, 0, ctypes.byref(nb), status.ctypes.data)
data = np.empty(int(nb.value), np.uint8); data[:] = 0
print("molecules", n, "cores", os.cpu_count())
for th in (1, 1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > 2 * (os.cpu_count() or 1): break
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); lib.pmx_pack_features(ctypes.byref(batch), th, offsets.ctypes.data, data.ctypes.data, data.size, ctypes.byref(nb), status.ctypes.data); best = min(best, time.perf_counter() - t0)
    print(f"{th:4d} threads {best:8.4f} s {n / best / 1e6:8.3f} M ligands/s", flush=True)
