#!/bin/bash
set -x
OUT=gpurun_out/r4c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tails.py tests/test_gpu_variants.py::test_arena_pass_is_retried_with_the_arena_empty -m gpu -q -s > $OUT/tails.log 2>&1; echo "tails rc=$?" >> $OUT/tails.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/bench_rb4.json 2> $OUT/bench_rb4.err
timeout 600 python tools/stress_shape.py > $OUT/stress_rb4.log 2>&1
PMX_CXXFLAGS=-DPMX_ROW_BATCH=8 python -m pharmaconet_amd.build --force > $OUT/build8.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/bench_rb8.json 2> $OUT/bench_rb8.err
timeout 600 python tools/stress_shape.py > $OUT/stress_rb8.log 2>&1
grep -h "subset pairs\|passed\|failed" $OUT/tails.log
grep -h "profiled pass" $OUT/bench_rb4.err $OUT/bench_rb8.err | cut -c1-200
tail -n 1 $OUT/stress_rb4.log $OUT/stress_rb8.log
