import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_golden
from pharmaconet_amd import PackedLibrary
from pharmaconet_amd.library import LigandFeatures, pack_ligand
model, lib, _, d = load_golden("set_c21_c8")
which = sys.argv[1]
if which in ("empty", "both"):
    empty = PackedLibrary.from_records([])
    print("screen empty...", flush=True)
    res = model.screen(empty, topk=3)
    print("done; ranking", res.ranking(), flush=True)
if which in ("g", "both"):
    zero = pack_ligand(LigandFeatures([6, 8], [[1], [0]], [], np.zeros((2, 4, 3), np.float32)))
    hal = pack_ligand(LigandFeatures([6, 17], [[1], [0]], [("Halogen", 1, 1)], np.ones((2, 4, 3), np.float32)))
    print("screen g...", flush=True)
    got = model.screen([zero, hal, lib.record(0)]).scores.cpu().numpy()
    print(got, flush=True)
