#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
B="python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg"
PMX_TREE_FLAGS=16384 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_tables_sq -o p -- $B > $OUT/pmc_tables_sq.log 2>&1
PMX_TREE_FLAGS=16384 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_tables_sq2 -o p -- $B > $OUT/pmc_tables_sq2.log 2>&1
PMX_TREE_FLAGS=16384 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/pmc_tables_tcp -o p -- $B > $OUT/pmc_tables_tcp.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_full_sq -o p -- $B > $OUT/pmc_full_sq.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/r4s/pmc_*')):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + '/*counter_collection.csv')
    if not fs:
        print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'ligand_kernel' in k or 'task_kernel' in k:
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    for k, v in acc.items():
        print(d.split('/')[-1], k, {c: round(x / 1e9, 3) for c, x in v.items()})
PY
tail -n 3 $OUT/pmc_tables_sq2.log $OUT/pmc_tables_tcp.log | cut -c1-300
