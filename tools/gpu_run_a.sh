#!/bin/bash
# round-4 GPU call A: tail parity tests + whole GPU suite + bench + instrumented counters
set -x
OUT=gpurun_out/r4a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tails.py -m gpu -x -q -s > $OUT/tails.log 2>&1; echo "tails rc=$?" >> $OUT/tails.log
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_tails.py > $OUT/gpu_tests.log 2>&1; echo "suite rc=$?" >> $OUT/gpu_tests.log
timeout 600 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
PMX_TREE_FLAGS=16384 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/bench_tables_only.json 2> $OUT/bench_tables_only.err
PMX_CXXFLAGS=-DPMX_COUNTERS python -m pharmaconet_amd.build --force > $OUT/build_counters.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/bench_counters.json 2> $OUT/bench_counters.err
tail -3 $OUT/tails.log $OUT/gpu_tests.log
