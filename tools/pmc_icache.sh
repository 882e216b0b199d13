#!/bin/bash
# Run on the GPU box: instruction-fetch / scalar-cache counters of one bench pass (separate --pmc runs, no tracing).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o -E "\b(SQ_[A-Z_0-9]*IFETCH[A-Z_0-9]*|SQC_[A-Z_0-9]+|SQ_INSTS_SMEM[A-Z_0-9]*|SQ_WAIT_INST_[A-Z_0-9]+|SQ_INST_CYCLES_[A-Z_0-9]+|SQ_BUSY_CYCLES|SQ_ACTIVE_INST_[A-Z_0-9]+|SQ_INSTS_BRANCH|SQ_INSTS_SENDMSG|SQ_INSTS_VSKIPPED|SQ_INSTS_FLAT[A-Z_0-9]*|SQ_INSTS_VMEM[A-Z_0-9]*)\b" $OUT/avail.txt | sort -u > $OUT/names.txt
cat $OUT/names.txt | tr '\n' ' '
run() { tag=$1; shift; timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAVE_CYCLES
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run cyc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES
python3 - <<'P'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_icache'
for tag in ('ic','dc','act','cyc'):
    for f in glob.glob(f'{out}/{tag}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-40:]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in acc.items():
            if 'ligand_kernel' in k or 'task_kernel' in k: print(tag,k,{a:round(b/1e9,3) for a,b in v.items()})
P
