#!/bin/bash
set -x
OUT=gpurun_out/r4f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/stress_shape.py > $OUT/stress.log 2>&1
PMX_TREE_FLAGS=16384 timeout 600 python tools/stress_shape.py > $OUT/stress_tables_only.log 2>&1
PMX_TREE_FLAGS=128 timeout 600 python tools/stress_shape.py > $OUT/stress_nofilter.log 2>&1
PMX_BUDGET=4096 timeout 600 python tools/stress_shape.py > $OUT/stress_budget4k.log 2>&1
PMX_WAVES_PER_CU=12 timeout 600 python tools/stress_shape.py > $OUT/stress_w12.log 2>&1
tail -n 1 $OUT/stress*.log
