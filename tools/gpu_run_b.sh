#!/bin/bash
set -x
OUT=gpurun_out/r4b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tails.py -m gpu -q -s > $OUT/tails.log 2>&1; echo "tails rc=$?" >> $OUT/tails.log
timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_tails.py > $OUT/gpu_tests.log 2>&1; echo "suite rc=$?" >> $OUT/gpu_tests.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python tools/stress_shape.py > $OUT/stress.log 2>&1
timeout 900 python tools/pockets16_breakdown.py > $OUT/pockets16.log 2>&1
tail -n 3 $OUT/tails.log $OUT/gpu_tests.log
