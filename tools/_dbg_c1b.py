import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from pharmaconet_amd import engine
from pharmaconet_amd.engine import screen
model, lib, weights, d = load_golden("set_6oim_c1")
ref = d["score"]
for li in (51, 13):
    for env in ({}, {"PMX_TREE_FLAGS": "2"}, {"PMX_BUDGET": "64"}, {"PMX_BUDGET": "64", "PMX_TREE_FLAGS": "4"}, {"PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0", "PMX_TREE_FLAGS": "4"}):
        for k, v in env.items(): os.environ[k] = v
        got = screen(model, lib, weights=weights, first=li, count=1).scores.cpu().numpy().astype(np.float64)
        st = engine.last_score_stats()
        for k in env: del os.environ[k]
        print(li, env, float(got[0]), float(ref[li]), {k: st[k] for k in ("n_steps", "n_iters", "n_tasks", "n_heavy", "max_iters_ligand")})
