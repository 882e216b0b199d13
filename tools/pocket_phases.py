"""Per-pocket phase shares of the 16 fixture pockets on the bench library (run on the GPU box). With a build made with
PMX_CXXFLAGS=-DPMX_TABLE_TICKS the `dbg` counters split the table phase: [0] self tables [1] centres of a level pair
[2] its node distances [3] prefilter and the rows of failing entries [4] items [5] chain lengths."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from pharmaconet_amd import PharmacophoreModel, engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
model = PharmacophoreModel.load(os.path.join(REPO, "tests", "golden", "model_6oim_like.pm"))
lib, offsets, data, _ = bench.build_library(model, n, 8, 4096, 0, "cuda")
pockets = [("6oim", model)] + [(f"p{k:02d}", PharmacophoreModel.load(os.path.join(REPO, "tests", "golden", "pockets16", f"model_{k:02d}.pm"))) for k in range(16)]
engine.set_profiling(True)
for name, m in pockets:
    m.screen(lib)
    torch.cuda.synchronize()
    t0 = time.time()
    m.screen(lib)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = engine.last_score_stats()
    alive = max(st["ticks_alive"], 1)
    sh = {k: round(st["ticks_" + k] / alive, 3) for k in ("scan", "tables", "bounds", "walk")}
    dbg = [round(x / alive, 3) for x in st["dbg"][:6]]
    if os.environ.get("PMX_RAW_DBG"):
        dbg = [st["dbg"][1], st["dbg"][5], round(st["dbg"][5] / max(st["dbg"][1] * 8, 1), 3)]
    print(f"{name} K={m.flat.num_clusters:2d} Nm={m.flat.num_nodes:2d} {dt*1e3:7.1f} ms  {len(lib)*8/dt/1e6:6.2f}M conf/s  shares {sh}  table parts {dbg}  "
          f"frames/lig {st['n_frames']/len(lib):.1f} passes/lig {st['n_passes']/len(lib):.1f} items/lig {st['n_items']/len(lib):.0f} probes {st['n_probe_passes']/len(lib):.1f}", flush=True)
