import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from pharmaconet_amd import engine
from pharmaconet_amd.engine import screen
model, lib, weights, d = load_golden("set_6oim_c1")
ref = d["score"]
for first, count in ((3, 1), (13, 1), (3, 11)):
    for env in ({}, {"PMX_BUDGET": "100", "PMX_MIN_LEVELS": "1"}):
        for k, v in env.items(): os.environ[k] = v
        got = screen(model, lib, weights=weights, first=first, count=count).scores.cpu().numpy().astype(np.float64)
        st = engine.last_score_stats()
        for k in env: del os.environ[k]
        err = np.abs(got - ref[first:first+count]) / np.maximum(np.abs(ref[first:first+count]), 1e-30)
        print(first, count, env, "bad", [(first + int(i), float(got[i]), float(ref[first + i])) for i in np.where(err > 1e-5)[0]], {k: st[k] for k in ("n_steps", "n_iters", "n_tasks", "n_heavy", "n_steps_first", "queue_overflow")})
