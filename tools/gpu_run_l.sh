#!/bin/bash
OUT=gpurun_out/r4l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -n 2 $OUT/tests.log
B="python bench.py --warmup 1 --steps 1 --no-cpu-baseline --no-serial-leg --pockets 16"
timeout 1500 $B --ligands 1253376 > $OUT/p16_shard.json 2> $OUT/p16_shard.err
PMX_OVERLAP=0 timeout 1500 $B --ligands 1253376 > $OUT/p16_shard_serial.json 2> $OUT/p16_shard_serial.err
timeout 600 $B --ligands 200000 > $OUT/p16_200k.json 2> $OUT/p16_200k.err
PMX_OVERLAP=0 timeout 600 $B --ligands 200000 > $OUT/p16_200k_serial.json 2> $OUT/p16_200k_serial.err
for f in $OUT/p16*.json; do python -c "import json; d=json.load(open('$f')); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],1))"; done
