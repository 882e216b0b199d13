import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_golden, rel_err
from pharmaconet_amd.engine import screen, last_score_stats
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else None
model, lib, weights, d = load_golden(name)
if n: lib = lib.slice(0, n)
t = time.time(); res = screen(model, lib, weights=weights); got = res.scores.cpu().numpy().astype(np.float64); dt = time.time() - t
ref = d["score"][: len(lib)]
err = rel_err(got[ref != 0], ref[ref != 0])
print(name, len(lib), f"{dt:.2f}s", "max rel", err.max() if len(err) else 0, "zero ok", np.all(got[ref == 0] == 0), last_score_stats())
