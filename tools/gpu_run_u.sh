#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4u
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_full_sq -o p -- $B > $OUT/pmc_full_sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_full_tcp -o p -- $B > $OUT/pmc_full_tcp.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/r4u/pmc_*')):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + '/*counter_collection.csv')
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'ligand_kernel' in k or 'task_kernel' in k:
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    for k, v in acc.items():
        print(d.split('/')[-1], k, {c: round(x / 1e9, 3) for c, x in v.items()})
PY
PMX_CXXFLAGS=-DPMX_COUNTERS=1 python -m pharmaconet_amd.build --force > $OUT/build_c1.log 2>&1
timeout 300 python bench.py --warmup 1 --steps 1 --no-cpu-baseline --no-serial-leg > $OUT/b_c1.json 2> $OUT/b_c1.err
grep "profiled pass" $OUT/b_c1.err | sed 's/.*n_probes/n_probes/'
