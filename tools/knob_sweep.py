#!/usr/bin/env python3
"""One resident bench library, many settings of libpmx's environment knobs (they are read per call): ms per pass, kernel times, work counters.

    python tools/knob_sweep.py [--ligands N] [--model 6oim|stress64] [--pockets P] "PMX_BUDGET=512" "PMX_TASK_DECAY_FROM=2 PMX_TASK_BUDGET_MIN=64" ...

An empty string is the default setting. Prints one line per setting (best of --reps timed passes)."""
import argparse
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ligands", type=int, default=0)
    ap.add_argument("--model", default="6oim")
    ap.add_argument("--pockets", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--library", choices=("expanded", "survey"), default="expanded")
    ap.add_argument("settings", nargs="*")
    args = ap.parse_args()
    import torch

    import bench
    import __graft_entry__ as entry

    entry.build()
    from pharmaconet_amd import PharmacophoreModel, engine

    model_file, conf, n_default, topo, active, seed = bench.WORKLOADS[args.model]
    model = PharmacophoreModel.load(REPO / "tests" / "golden" / model_file)
    pockets = [model]
    if args.pockets > 1:
        pockets = [PharmacophoreModel.load(REPO / "tests" / "golden" / "pockets16" / f"model_{k:02d}.pm") for k in range(args.pockets)]
    device = torch.device("cuda", 0)
    if args.library == "survey":
        lib, offsets, data, _ = bench.build_survey_library(model, args.ligands or n_default, conf, 0, device, active)
    else:
        lib, offsets, data, _ = bench.build_library(model, args.ligands or n_default, conf, topo, 0, device, active, seed)
    n_conf = lib.total_conformers * len(pockets)
    ref = None
    for setting in [""] + list(args.settings):
        saved = {}
        for kv in setting.split():
            k, v = kv.split("=", 1)
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            engine.release_workspaces()  # (buffer sizes are knobs too)
            for pocket in pockets:
                engine.screen(pocket, lib, topk=1000)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(args.reps):
                t0 = time.perf_counter()
                for pocket in pockets:
                    res = engine.screen(pocket, lib, topk=1000)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            engine.set_profiling(True)
            res = engine.screen(pockets[-1], lib, topk=1000)
            torch.cuda.synchronize()
            st = engine.last_score_stats()
            engine.set_profiling(False)
            same = "first" if ref is None else ("same bits" if torch.equal(res.scores, ref) else "SCORES DIFFER")
            if ref is None:
                ref = res.scores.clone()
            n = len(lib)
            print(f"{setting or '(default)':60s} {best * 1e3:8.2f} ms {n_conf / best / 1e6:7.2f} M/s | lig {st['ms_ligand']:6.1f} tasks {st['ms_tasks']:5.1f} | "
                  f"frames {st['n_frames'] / n:6.1f} passes {st['n_passes'] / n:6.1f} tasks/lig {st['n_tasks'] / n:5.2f} over {st['n_heavy'] / n:5.3f} max {st['max_passes']} "
                  f"qovf {st['queue_overflow']} arena {st['arena_bytes'] / 2**30:.1f}/{st.get('arena_capacity', 0) / 2**30:.0f} GB | {same}", flush=True)
            if any(st.get("dbg", [])):  # instrumented builds (-DPMX_COUNTERS / _TICKS): the walker's counters per ligand
                print("    dbg/ligand:", [round(x / n, 2) for x in st["dbg"]], "path tests", round(st.get("n_path_bounds", 0) / n, 1), "drops", round(st.get("n_path_drops", 0) / n, 1), flush=True)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


if __name__ == "__main__":
    main()
