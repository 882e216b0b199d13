#!/usr/bin/env python3
"""Print per-launch durations of the pmx kernels from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "pmx::" in n:
        print(n.split("(")[0][-48:], "grid", r.get("Grid_Size", ""), "wg", r.get("Workgroup_Size", ""), "lds", r.get("LDS_Block_Size", ""),
              "vgpr", r.get("VGPR_Count", ""), "%.3f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
