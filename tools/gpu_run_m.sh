#!/bin/bash
OUT=gpurun_out/r4m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PMX_WRITE_PROFILES=$OUT timeout 900 python -m pytest tests/test_gpu_variants.py::test_stress_config_at_full_size -m gpu -q -x > $OUT/stress_test.log 2>&1; tail -n 2 $OUT/stress_test.log
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/b_1M.json 2> $OUT/b_1M.err
python -c "import json; d=json.load(open('$OUT/b_1M.json')); print('bench', round(d['value']/1e6,3), round(d['ms_per_step'],1)); print(d['end_to_end'])"
timeout 600 python tools/stress_shape.py 196 > $OUT/stress64.log 2>&1; tail -n 1 $OUT/stress64.log | cut -c1-60
