cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_sq_x; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg > $OUT/sq.log 2>&1
python - <<'P'
import csv,glob,collections,os
OUT=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_sq_x'
fs=glob.glob(OUT+'/sq/**/*counter_collection.csv',recursive=True)
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name'].split('(')[0][-30:]
    tot[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in tot.items():
    if 'ligand_kernel' in k or 'task_kernel' in k:
        print(k,{a:round(b/2/200704) for a,b in v.items()})
P
