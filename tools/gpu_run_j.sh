#!/bin/bash
OUT=gpurun_out/r4j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -n 2 $OUT/tests.log
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
timeout 300 $B > $OUT/b.json 2> $OUT/b.err
python -c "import json; d=json.load(open('$OUT/b.json')); print('bench', round(d['value']/1e6,3), round(d['ms_per_step'],1), d['roofline']['kernel_ms_per_launch'])"
timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/p16.json 2> $OUT/p16.err
python -c "import json; d=json.load(open('$OUT/p16.json')); print('p16', round(d['value']/1e6,3), round(d['ms_per_step'],1))"
timeout 600 python tools/stress_shape.py > $OUT/stress.log 2>&1; tail -n 1 $OUT/stress.log | cut -c1-40
