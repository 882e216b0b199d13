#!/bin/bash
# Run on the GPU box: vector memory pipeline (TA / TCP / TD) counters of one bench pass and of the tables alone (PMX_TREE_FLAGS=16384); --pmc only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_tcp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; }
for mode in full tabs; do
  if [ $mode = tabs ]; then export PMX_TREE_FLAGS=16384; else unset PMX_TREE_FLAGS; fi
  run ${mode}_ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WAVEFRONTS GRBM_GUI_ACTIVE
  run ${mode}_tcp1 TCP_GATE_EN1 TCP_GATE_EN2 TCP_TOTAL_CACHE_ACCESSES TCP_TOTAL_ACCESSES
  run ${mode}_tcp2 TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_PENDING_STALL_CYCLES
  run ${mode}_tcp3 TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_WRITE_REQ
  run ${mode}_tlb TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TD_TD_BUSY
done
python3 - <<'P'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_tcp'
for d in sorted(glob.glob(out+'/*/')):
    tag=os.path.basename(d.rstrip('/'))
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-34:]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in acc.items():
            if 'ligand_kernel' in k or 'task_kernel' in k: print(tag,k,{a:round(b/1e9,4) for a,b in v.items()})
P
