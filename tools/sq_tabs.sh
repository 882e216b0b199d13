ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/sq_tabs; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; }
PMX_TREE_FLAGS=16384 run tabs SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run clk GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU
python3 - <<'P'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sq_tabs'
for tag in ('tabs','clk'):
    for f in glob.glob(f'{out}/{tag}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-34:]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
            if r['Counter_Name']=='SQ_INSTS_SALU': n[k]+=1
        for k,v in acc.items():
            if 'ligand_kernel' in k or 'task_kernel' in k: print(tag,k,n[k],{a:round(b/1e9,4) for a,b in v.items()})
    for f in glob.glob(f'{out}/{tag}/**/*kernel_trace.csv', recursive=True):
        d=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-34:]
            d[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        print(tag,'ms',{k:round(v,2) for k,v in d.items() if 'ligand' in k or 'task' in k})
P
