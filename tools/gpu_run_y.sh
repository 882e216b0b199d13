#!/bin/bash
OUT=gpurun_out/r4y
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/b_$name.json 2> $OUT/b_$name.err; python -c "import json; d=json.load(open('$OUT/b_$name.json')); print('$name', round(d['value']/1e6,3), round(d['ms_per_step'],1), [round(x,1) for x in d['roofline']['kernel_ms_per_launch'].values()])"; }
for IB in 3 4; do
  PMX_CXXFLAGS="-DPMX_ITEM_BATCH=$IB" python -m pharmaconet_amd.build --force > $OUT/build_$IB.log 2>&1
  run ib$IB X=1
  run ib${IB}_tables PMX_TREE_FLAGS=16384
done
PMX_CXXFLAGS="-DPMX_SCREEN_WAVES=7" python -m pharmaconet_amd.build --force > $OUT/build_w7.log 2>&1
run w7 X=1
run w7_tables PMX_TREE_FLAGS=16384
