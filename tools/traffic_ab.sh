#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/traffic_ab.sh [variant ...]'): L2 <-> fabric bytes per ligand of the engine's kernels (rocprofv3 --pmc
# FETCH_SIZE / WRITE_SIZE, separate passes, read side doubled as MI355X_MICROARCH.md prescribes) for the product build and variants/libpmx_<name>.so.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in product "$@"; do
  if [ "$name" = product ]; then unset PMX_LIBPMX; else export PMX_LIBPMX=$ROOT/variants/libpmx_$name.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${name}_$c
    rocprofv3 --pmc $c --output-format csv -d $OUT/${name}_$c -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 ${TRAFFIC_ARGS:-} --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/${name}_$c.log 2>&1
  done
  python - "$OUT" "$name" <<'PY'
import collections, csv, sys
from pathlib import Path
out, name = Path(sys.argv[1]), sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = next((out / f"{name}_{ctr}").rglob("*counter_collection.csv"))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "pmx::" in k and row["Counter_Name"] == ctr:
            short = "ligand_kernel" if "ligand_kernel" in k else ("task_kernel" if "task_kernel" in k else None)
            if short:
                tot[short][ctr] += float(row["Counter_Value"])
                if ctr == "FETCH_SIZE": launches[short] += 1
passes = max(1, launches["ligand_kernel"] // 3)
n = 200704 * passes
line = []
for k in ("ligand_kernel", "task_kernel"):
    f, w = tot[k]["FETCH_SIZE"], tot[k]["WRITE_SIZE"]
    line.append(f"{k}: read {2 * f * 1024 / n / 1e3:.1f} KB write {w * 1024 / n / 1e3:.1f} KB")
print(f"{name}: {passes} passes | " + " | ".join(line) + f" | total {sum((2 * tot[k]['FETCH_SIZE'] + tot[k]['WRITE_SIZE']) for k in tot) * 1024 / n / 1e3:.1f} KB per ligand")
PY
done
