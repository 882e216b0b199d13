import sys, os, time; sys.path.insert(0, ".")
import torch, bench
from pharmaconet_amd import PharmacophoreModel, engine
m = PharmacophoreModel.load("tests/golden/model_6oim_like.pm")
lib, *_ = bench.build_library(m, 1000000, 8, 4096, 0, torch.device("cuda", 0), 0.1, (None, 0))
for _ in range(2): r = engine.screen(m, lib, topk=10); torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); r = engine.screen(m, lib, topk=10); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
st = engine.last_score_stats()
print("MAXEXP", os.environ.get("PMX_FN_MAXEXP"), "ms", round(best * 1e3, 2), "exact self values/ligand", st["n_exact_values"] / len(lib), "checksum", float(r.scores.double().sum()))
