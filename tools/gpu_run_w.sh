#!/bin/bash
OUT=gpurun_out/r4w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/b_$name.json 2> $OUT/b_$name.err; python -c "import json; d=json.load(open('$OUT/b_$name.json')); print('$name', round(d['value']/1e6,3), round(d['ms_per_step'],1), [round(x,1) for x in d['roofline']['kernel_ms_per_launch'].values()])"; }
run base X=1
run s512k_50 PMX_SUPER=524288
run s512k_75 PMX_SUPER=524288 PMX_LIG_SHARE=0.75
run s512k_85 PMX_SUPER=524288 PMX_LIG_SHARE=0.85
run s256k_75 PMX_SUPER=262144 PMX_OVERLAP=2 PMX_LIG_SHARE=0.75
run s256k_85 PMX_SUPER=262144 PMX_OVERLAP=2 PMX_LIG_SHARE=0.85
run s128k_85 PMX_SUPER=131072 PMX_OVERLAP=2 PMX_LIG_SHARE=0.85
run tb2048 PMX_TASK_BUDGET=2048
run tb2048r4 PMX_TASK_BUDGET=2048 PMX_ROUNDS=4
run r6 PMX_ROUNDS=6
run b768 PMX_BUDGET=768 PMX_TASK_BUDGET=4096 PMX_ROUNDS=4
