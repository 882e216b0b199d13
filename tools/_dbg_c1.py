import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden, rel_err
from pharmaconet_amd.engine import screen
name = sys.argv[1] if len(sys.argv) > 1 else "set_6oim_c1"
model, lib, weights, d = load_golden(name)
ref = d["score"]
for env in ({}, {"PMX_TREE_FLAGS": "16"}, {"PMX_TREE_FLAGS": "20"}):
    for k, v in env.items(): os.environ[k] = v
    got = screen(model, lib, weights=weights).scores.cpu().numpy().astype(np.float64)
    for k in env: del os.environ[k]
    nz = ref != 0
    err = np.zeros_like(ref); err[nz] = np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])
    bad = np.where(err > 2e-6)[0]
    print(env, "max err %.3e" % err.max(), "bad", bad[:10], [(float(got[i]), float(ref[i])) for i in bad[:4]])
