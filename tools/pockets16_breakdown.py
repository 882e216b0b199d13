import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pathlib import Path
from pharmaconet_amd import PharmacophoreModel, engine
REPO = Path(__file__).resolve().parent.parent
model = PharmacophoreModel.load(REPO / "tests" / "golden" / "model_6oim_like.pm")
lib, offsets, data, molecules = bench.build_library(model, 200000, 8, 4096, 0, torch.device("cuda", 0))
engine.set_profiling(True)
tot = 0
for k in range(16):
    m = PharmacophoreModel.load(REPO / "tests" / "golden" / "pockets16" / f"model_{k:02d}.pm")
    engine.screen(m, lib, topk=1000); torch.cuda.synchronize()
    t0 = time.time(); engine.screen(m, lib, topk=1000); torch.cuda.synchronize(); dt = time.time() - t0
    st = engine.last_score_stats(); tot += dt
    n = len(lib)
    print(f"pocket {k:2d} K={m.flat.num_clusters:2d} Nm={m.flat.num_nodes:2d} {dt*1e3:7.1f} ms  lig {st['ms_ligand']:6.1f} task {st['ms_tasks']:6.1f} frames/l {st['n_frames']/n:8.0f} passes/l {st['n_passes']/n:8.0f} items/lc {st['n_items']/n/8:7.0f} tasks/l {st['n_tasks']/n:6.1f} heavy {st['n_heavy']} sovf {st['n_slice_overflow']} qovf {st['queue_overflow']} maxp {st['max_passes']} share tables {st['ticks_tables']/max(st['ticks_alive'],1):.2f} bounds {st['ticks_bounds']/max(st['ticks_alive'],1):.2f} walk {st['ticks_walk']/max(st['ticks_alive'],1):.2f}")
print("total", tot, "->", 16*len(lib)*8/tot/1e6, "M")
