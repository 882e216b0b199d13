"""Repeatability soak: the same 1M-ligand pass several times; every run must give the same bits."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from pharmaconet_amd.constants import TYPE_ID  # noqa: E402
from pharmaconet_amd.engine import DeviceLibrary  # noqa: E402
from tools.synthetic import BASE_SEED, expand_library_on_device, synthetic_library  # noqa: E402

model, _, _, _ = load_golden("set_6oim_c8")
st = model.__getstate__()
centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
base = synthetic_library(4096, first=0, num_conformers=8, model_nodes=(centers, types), active_fraction=0.1,
                         seed=BASE_SEED, max_nodes=32, conformer_noise=0.0)
offsets, data = expand_library_on_device(base, 245, "cuda", seed=BASE_SEED)
lib = DeviceLibrary.from_device_buffers(offsets, data)
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    s = model.screen(lib, topk=1000)
    sc = s.scores
    assert torch.isfinite(sc).all()
    if ref is None:
        ref, ref_top = sc.clone(), s.topk_indices.clone()
    else:
        assert torch.equal(sc, ref), f"run {it}: scores differ at {(sc != ref).nonzero().flatten()[:5].tolist()}"
        assert torch.equal(s.topk_indices, ref_top), f"run {it}: top-k differs"
print("soak ok:", float(ref.double().sum()))
