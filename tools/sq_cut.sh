#!/bin/bash
# Run on the GPU box: instruction counts (SQ_INSTS_*) of the table phase alone (PMX_TREE_FLAGS=16384) for the product build and for analysis builds that leave a
# section out (variants/libpmx_cut*.so, -DPMX_CUT): the difference is the section's instruction budget.
ROOT=${GRAFT_REPO_ROOT:-.}; OUT=$ROOT/gpurun_out/sq_cut; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for name in product "$@"; do
  if [ "$name" = product ]; then unset PMX_LIBPMX; else export PMX_LIBPMX=$ROOT/variants/libpmx_$name.so; fi
  rm -rf $OUT/$name
  PMX_TREE_FLAGS=16384 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/$name -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/$name.log 2>&1
  python3 - "$OUT/$name" "$name" <<'P'
import csv, glob, collections, sys
acc = collections.defaultdict(float); n = 0
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ligand_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
            n += r['Counter_Name'] == 'SQ_INSTS_VALU'
per = 200704 * max(1, n // 3)
print(sys.argv[2], {k.replace('SQ_INSTS_', ''): round(v / per) for k, v in sorted(acc.items())})
P
done
