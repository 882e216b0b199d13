#!/bin/bash
set -x
OUT=gpurun_out/r4g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
for IB in 2 3 4; do
  PMX_CXXFLAGS="-DPMX_ITEM_BATCH=$IB" python -m pharmaconet_amd.build --force > $OUT/build_$IB.log 2>&1
  timeout 300 $B > $OUT/b_ib$IB.json 2> $OUT/b_ib$IB.err
  PMX_TREE_FLAGS=16384 timeout 300 $B > $OUT/b_ib${IB}_tables.json 2> $OUT/b_ib${IB}_tables.err
  timeout 600 python tools/stress_shape.py > $OUT/stress_ib$IB.log 2>&1
  PMX_TREE_FLAGS=16384 timeout 600 python tools/stress_shape.py > $OUT/stress_ib${IB}_tables.log 2>&1
done
tail -n 2 $OUT/tests.log
for f in $OUT/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value']/1e6, d['ms_per_step'])"; done
tail -qn 1 $OUT/stress_ib*.log | cut -c1-60
