import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden
from pharmaconet_amd import engine
from pharmaconet_amd.constants import TYPE_ID
from pharmaconet_amd.engine import DeviceLibrary
from tools.synthetic import expand_library_on_device, synthetic_library
model, _, _, _ = load_golden("set_s64_c64")
st = model.__getstate__()
centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
base = synthetic_library(512, num_conformers=64, model_nodes=(centers, types), active_fraction=0.2, seed=6464, max_nodes=32, conformer_noise=0.0)
offsets, data = expand_library_on_device(base, int(sys.argv[1]) if len(sys.argv) > 1 else 40, "cuda", seed=6465)
lib = DeviceLibrary.from_device_buffers(offsets, data)
print("ligands", len(lib), "K", model.flat.num_clusters, "Nm", model.flat.num_nodes)
engine.set_profiling(True)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    s = model.screen(lib).scores
    torch.cuda.synchronize(); dt = time.time() - t0
    stt = engine.last_score_stats()
    print(f"{dt*1e3:.1f} ms  {len(lib)*64/dt/1e6:.2f} M conf/s", {k: stt[k] for k in ("ms_ligand", "ms_tasks", "n_frames", "n_passes", "n_items", "n_slice_overflow", "n_heavy", "n_tasks", "max_passes", "arena_bytes", "n_probes", "n_probe_passes")},
          {k: round(stt["ticks_" + k] / max(stt["ticks_alive"], 1), 3) for k in ("scan", "tables", "bounds", "walk")})
