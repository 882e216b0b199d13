#!/bin/bash
OUT=gpurun_out/r4i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --steps 1 --no-cpu-baseline --no-serial-leg --pockets 16 --ligands 200000"
run() { name=$1; shift; env "$@" timeout 600 $B > $OUT/p16_$name.json 2> $OUT/p16_$name.err; python -c "import json; d=json.load(open('$OUT/p16_$name.json')); print('$name', round(d['value']/1e6,3), round(d['ms_per_step'],1))"; }
run base X=1
run budget256 PMX_BUDGET=256
run budget1024 PMX_BUDGET=1024
run budget2048 PMX_BUDGET=2048
run bc0 PMX_BOUND_COST=0
run bc32k PMX_BOUND_COST=32768
run bc128k PMX_BOUND_COST=131072
run slice256 PMX_SLICE_KB=256
run slice128 PMX_SLICE_KB=128
run w16 PMX_WAVES_PER_CU=16
run minlev2 PMX_MIN_LEVELS=2
run minlev4 PMX_MIN_LEVELS=4
run rounds6 PMX_ROUNDS=6
run taskb256 PMX_TASK_BUDGET=256
run taskb2048 PMX_TASK_BUDGET=2048
