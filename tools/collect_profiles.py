#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one gpurun call into the tracked summaries under profiles/.

  python tools/collect_profiles.py <bench.json> <kernel_stats.csv> <fetch_dir> <write_dir> <sq_dir> [ligands_in_pmc_run]

The PMC passes are separate runs (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, one SQ pass) of
`bench.py --ligands 200000 --steps 1 --warmup 0`; FETCH_SIZE / WRITE_SIZE are KiB and the read side is
doubled as MI355X_MICROARCH.md prescribes for gfx950.
"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

import os

REPO = Path(__file__).resolve().parents[1]
PROF = REPO / "profiles"
TAG = os.environ.get("PMX_PROFILE_TAG", "r6")  # file name prefix: the round the profiles belong to


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    bench, kstats, fetch_dir, write_dir, sq_dir = (Path(a) for a in sys.argv[1:6])
    n_lig = int(sys.argv[6]) if len(sys.argv) > 6 else 200704
    # the build the counters belong to: tools/profile_round.sh leaves bench.csrc_digest() of the tree it ran on next to them; bench.py quotes
    # the summaries only while csrc/ still hashes to it
    sha_file = bench.parent / "csrc_sha16.txt"
    if sha_file.exists():
        sha = sha_file.read_text().strip()
    else:
        sys.path.insert(0, str(REPO))
        import bench as bench_module

        sha = bench_module.csrc_digest()
    PROF.mkdir(exist_ok=True)
    shutil.copy(bench, PROF / f"{TAG}_bench_1M.json")
    # kernel stats: keep pmx kernels and the five largest others
    rows = list(csv.reader(open(kstats)))
    keep = [rows[0]] + [r for r in rows[1:] if "pmx::" in r[0] or "cub" in r[0].lower()]
    csv.writer(open(PROF / f"{TAG}_kernel_stats.csv", "w", newline="")).writerows(keep)
    res = {}
    for d, ctr, out in ((fetch_dir, "FETCH_SIZE", f"{TAG}_pmc_fetch_size.csv"), (write_dir, "WRITE_SIZE", f"{TAG}_pmc_write_size.csv")):
        f = next(d.glob("*counter_collection.csv"))
        rows = list(csv.reader(open(f)))
        ki = rows[0].index("Kernel_Name")
        csv.writer(open(PROF / out, "w", newline="")).writerows([rows[0]] + [r for r in rows[1:] if "pmx::" in r[ki]])
        tot, cnt = collections.defaultdict(float), collections.Counter()
        for row in csv.DictReader(open(f)):
            if "pmx::" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                k = short(row["Kernel_Name"])
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
        res[ctr] = (tot, cnt)
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --ligands 200000 --steps 1 --warmup 0",
        "note": "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_ligand = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / (ligands * passes) for the engine's kernels "
                "(`passes` engine passes ran under the profiler: the timed step and bench.py's profiled pass; the summaries of rounds 3 "
                "and of the first round-4 sets divided by ligands only and are 2x too high), the read side doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; narrow accesses are "
                "uncalibrated, so the read figure is an upper estimate). The counters see L2 <-> fabric traffic, i.e. they "
                "include what the 256 MB Infinity Cache serves (the per-wavefront table slices).",
        "ligands": n_lig,
        "csrc_sha16": sha,
        "kernels": {},
    }
    # passes of the engine in the profiled command: bench.py makes its timed steps and one more pass for the kernel times (3 ligand-kernel
    # launches per pass) - counters are summed over all of them, so per-ligand figures divide by ligands x passes
    n_pass = max(1, max((c for k, c in res["FETCH_SIZE"][1].items() if "ligand_kernel" in k), default=3) // 3)
    out["passes"] = n_pass
    for k, f in res["FETCH_SIZE"][0].items():
        w = res["WRITE_SIZE"][0].get(k, 0.0)
        if f + w < 1000:
            continue
        out["kernels"][k] = {
            "fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_ligand": (2 * f + w) * 1024 / (n_lig * (n_pass if "pmx::ligand_kernel" in k or "pmx::task_kernel" in k or "finalize" in k or "round_kernel" in k else 1)),
            "launches": res["FETCH_SIZE"][1][k],
        }
        for short_name in ("ligand_kernel", "task_kernel"):  # the keys bench.py looks up
            if short_name in k:
                out["kernels"][short_name] = out["kernels"][k]
    json.dump(out, open(PROF / f"{TAG}_hbm_traffic.json", "w"), indent=1)
    sq = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(next(sq_dir.glob("*counter_collection.csv")))):
        k = row["Kernel_Name"]
        if "pmx::" in k and ("ligand_kernel" in k or "task_kernel" in k):
            sq[short(k).replace("pmx::", "")][row["Counter_Name"]] += float(row["Counter_Value"])
    clock = None
    grbm_dir = sq_dir.parent / "pmc_GRBM"  # (tools/profile_round.sh; absent in older sets)
    if grbm_dir.is_dir():
        cyc = ms = 0.0
        for row in csv.DictReader(open(next(grbm_dir.glob("*counter_collection.csv")))):
            if "pmx::ligand_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cyc += float(row["Counter_Value"])
        for row in csv.DictReader(open(next(grbm_dir.glob("*kernel_trace.csv")))):
            if "pmx::ligand_kernel" in row["Kernel_Name"]:
                ms += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        if cyc and ms:
            clock = {"GRBM_GUI_ACTIVE": cyc, "kernel_ms": ms, "kernel": "ligand_kernel launches of the same command under --pmc GRBM_GUI_ACTIVE --kernel-trace", "xcds": 8,
                     "clock_hz": cyc / 8 / (ms * 1e-3), "note": "GRBM_GUI_ACTIVE summed over the 8 XCDs / kernel time: an upper estimate of the shader clock under this load"}
    json.dump({**({"clock": clock} if clock else {}), "source": "rocprofv3 --pmc SQ_* (one profiler run) of bench.py --ligands 200000 --steps 1 --warmup 0; SQ_WAVE_CYCLES, "
                         "SQ_WAIT_* and SQ_ACTIVE_* count quad-cycles; totals over the run's engine passes (SQ_WAVES / 18432 = passes)",
               "ligands": n_lig, "passes": n_pass, "csrc_sha16": sha, "counters": sq},
              open(PROF / f"{TAG}_pmc_sq_summary.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:40s} fetch {v['fetch_kib_raw'] / 1e6:8.3f} GiB  write {v['write_kib'] / 1e6:8.3f} GiB  {v['hbm_bytes_per_ligand']:10.0f} B/ligand")
    for k, v in sq.items():
        print(k, {c: round(x / 1e9, 2) for c, x in v.items()})


if __name__ == "__main__":
    main()
