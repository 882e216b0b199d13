#!/bin/bash
OUT=gpurun_out/r4full
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "suite rc=$?" >> $OUT/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
tail -n 5 $OUT/gpu_tests.log; tail -n 2 $OUT/smoke.log
