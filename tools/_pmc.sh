cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_r3a; mkdir -p $OUT
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg > $OUT/$1.log 2>&1; }
run sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python - <<'P'
import csv,glob,collections,os
OUT=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_r3a'
for d in ("sq","sq2","tcc","tcp","fetch","write"):
    fs=glob.glob(OUT+'/'+d+'/**/*counter_collection.csv',recursive=True)
    if not fs: print(d,"no output"); os.system("tail -3 "+OUT+"/"+d+".log"); continue
    tot=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0][-40:]
        tot[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in tot.items():
        if 'ligand_kernel' in k or 'task_kernel' in k:
            print(d,k,dict(v))
P
