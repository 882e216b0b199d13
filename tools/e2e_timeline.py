#!/usr/bin/env python3
"""Timeline of the last device-packed end-to-end run from a rocprofv3 trace (kernel + memory-copy CSVs of `bench.py`):

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python bench.py --no-survey-leg --no-cpu-baseline --no-parity-sample
    python tools/e2e_timeline.py DIR/t

Prints, for the last window that holds three record_kernel launches, every kernel family and copy direction with first start / last end (ms from the window's start) and busy time."""
import csv
import sys
from collections import defaultdict


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def main():
    base = sys.argv[1]
    k = rows(base + "_kernel_trace.csv")
    try:
        m = rows(base + "_memory_copy_trace.csv")
    except FileNotFoundError:
        m = []
    ev = []
    for r in k:
        name = r["Kernel_Name"]
        short = next((s for s in ("graph_wave_kernel", "graph_kernel", "record_kernel", "ligand_kernel", "task_kernel", "finalize_kernel", "library_stats", "topk", "scan") if s in name), name[:40])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "?"))))
    for r in m:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?"), "-"))
    ev.sort()
    rec = [e for e in ev if e[2] == "record_kernel"]
    if len(rec) < 3:
        sys.exit("no device-packed run in the trace")
    last3 = rec[-4:-1] if len(rec) >= 4 else rec[-3:]  # (the very last record_kernel is the packer-alone timing after the runs)
    t_lo = last3[0][0] - 30_000_000
    t_hi = last3[-1][1] + 150_000_000
    win = [e for e in ev if t_lo <= e[0] <= t_hi]
    t0 = min(e[0] for e in win if e[2].startswith("copy:") and e[0] >= last3[0][0] - 20_000_000) if any(e[2].startswith("copy:") for e in win) else win[0][0]
    print(f"window from {t0} ns")
    for e in win:
        if e[0] < t0:
            continue
        if e[2] in ("task_kernel", "finalize_kernel", "scan") or e[1] - e[0] < 300_000 and not e[2].startswith(("graph", "record")):
            continue
        print(f"{(e[0] - t0) / 1e6:9.2f} .. {(e[1] - t0) / 1e6:9.2f} ms  {(e[1] - e[0]) / 1e6:8.2f}  {e[2]:22s} stream {e[3]}")
    busy = defaultdict(float)
    for e in win:
        if e[0] >= t0:
            busy[e[2]] += (e[1] - e[0]) / 1e6
    print({k_: round(v, 2) for k_, v in sorted(busy.items(), key=lambda kv: -kv[1])[:12]})


if __name__ == "__main__":
    main()
