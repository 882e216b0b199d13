cd /tmp && export TMPDIR=/tmp
N=${1:-200000}
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3b -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --ligands $N --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg > $GRAFT_REPO_ROOT/gpurun_out/prof_r3b.log 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/prof_r3b && python - <<'P'
import csv,glob
f=glob.glob('**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last pass: from the last ctl_clear_kernel
idx=[i for i,r in enumerate(rows) if 'ctl_clear' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if 'rocprim' in r['Kernel_Name'] or 'topk' in r['Kernel_Name'] or 'copyBuffer' in r['Kernel_Name']: continue
    print("%10.3f %9.3f ms  %s"%((int(r['Start_Timestamp'])-t0)/1e6,d,r['Kernel_Name'][:50]))
P
