#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/gpu_suite.sh): the whole -m gpu suite under time limits, the smoke entry, one bench line.
OUT=gpurun_out/suite
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x --durations=15 > $OUT/gpu_tests.log 2>&1; echo "suite rc=$?" >> $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
grep "passed\|failed\|rc=" $OUT/gpu_tests.log | tail -n 3; tail -n 2 $OUT/smoke.log
timeout 600 python bench.py --warmup 1 --steps 3 > $OUT/b.json 2> $OUT/b.err
python -c "import json; d=json.load(open('$OUT/b.json')); print('bench', round(d['value']/1e6,3), round(d['ms_per_step'],1), [round(x,1) for x in d['roofline']['kernel_ms_per_launch'].values()], d['parity_sample'], d['cpu_baseline'])"
