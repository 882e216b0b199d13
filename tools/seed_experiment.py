import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as e; e.build()
from pharmaconet_amd import PharmacophoreModel, engine
from pharmaconet_amd.constants import TYPE_ID
from pharmaconet_amd.engine import DeviceLibrary
from pharmaconet_amd.synthetic import BASE_SEED, expand_library_on_device, synthetic_library
model = PharmacophoreModel.load("tests/golden/model_6oim_like.pm")
st = model.__getstate__()
centers = np.array([n["center"] for n in st["nodes"]]); types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
base = synthetic_library(4096, num_conformers=8, model_nodes=(centers, types), active_fraction=0.1, seed=BASE_SEED, max_nodes=32, conformer_noise=0.0)
off, dat = expand_library_on_device(base, 49, "cuda", seed=BASE_SEED)
lib = DeviceLibrary.from_device_buffers(off, dat)
os.environ["PMX_BUDGET"] = "100000000"; os.environ["PMX_COOP_KB"] = "400"
os.environ["PMX_SEED_BEST"] = "1"
r1 = engine.screen(model, lib).scores.clone(); s1 = engine.last_score_stats()["n_steps"]
os.environ["PMX_SEED_BEST"] = "2"
r2 = engine.screen(model, lib).scores.clone(); s2 = engine.last_score_stats()["n_steps"]
print("ligands", len(lib), "steps plain", s1, "steps seeded with the final maxima", s2, "ratio", s1 / max(s2, 1), "equal", bool(torch.equal(r1, r2)))
