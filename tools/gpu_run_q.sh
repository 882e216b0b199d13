#!/bin/bash
OUT=gpurun_out/r4q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/b_$name.json 2> $OUT/b_$name.err; python -c "import json; d=json.load(open('$OUT/b_$name.json')); print('$name', round(d['value']/1e6,3), round(d['ms_per_step'],1), [round(x,1) for x in d['roofline']['kernel_ms_per_launch'].values()], round(d['work']['tree_frames_per_ligand'],1), round(d['work']['walker_passes_per_ligand'],1), d['work']['wave_time_share'])"; }
run base X=1
run tables PMX_TREE_FLAGS=16384
run budget256 PMX_BUDGET=256
run budget1024 PMX_BUDGET=1024
run budget2048 PMX_BUDGET=2048
run rounds6 PMX_ROUNDS=6
run bc0 PMX_BOUND_COST=0
for ML in 2 4; do
  PMX_CXXFLAGS="-DPMX_PATH_MIN_LEVELS=$ML" python -m pharmaconet_amd.build --force > $OUT/build_ml$ML.log 2>&1
  run minlev$ML X=1
done
PMX_CXXFLAGS="-DPMX_TC_LEVELS=2" python -m pharmaconet_amd.build --force > $OUT/build_tc2.log 2>&1
run tc2 X=1
