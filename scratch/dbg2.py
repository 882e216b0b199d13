import sys, numpy as np, subprocess, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
case = sys.argv[1]
from conftest import load_golden
from pharmaconet_amd import PackedLibrary
from pharmaconet_amd.library import LigandFeatures, pack_ligand
model, lib, _, d = load_golden("set_c21_c8")
zero = pack_ligand(LigandFeatures([6, 8], [[1], [0]], [], np.zeros((2, 4, 3), np.float32)))
zero8 = pack_ligand(LigandFeatures([6, 8], [[1], [0]], [], np.zeros((2, 8, 3), np.float32)))
hal = pack_ligand(LigandFeatures([6, 17], [[1], [0]], [("Halogen", 1, 1)], np.ones((2, 4, 3), np.float32)))
cases = {"a": [lib.record(0)], "b": [zero, lib.record(0)], "c": [lib.record(0), zero], "d": [zero8, lib.record(0)],
         "e": [lib.record(0), lib.record(1)], "f": [lib.record(1)], "g": [zero, hal, lib.record(0)]}
got = model.screen(cases[case]).scores.cpu().numpy()
print(case, got, d["score"][:2])
