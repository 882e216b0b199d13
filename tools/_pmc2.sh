cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_r3b; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg > $OUT/$1.log 2>&1; }
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"
run fetch "FETCH_SIZE"
python - <<'P'
import csv,glob,collections,os
OUT=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_r3b'
for d in ("tcc","fetch"):
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
