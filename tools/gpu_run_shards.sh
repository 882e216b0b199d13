#!/bin/bash
OUT=gpurun_out/r4shards2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 2 --warmup 1 --ligands 12500000 --no-cpu-baseline --no-serial-leg > $OUT/b_shard12M.json 2> $OUT/b_shard12M.err
PMX_OVERLAP=0 timeout 600 python bench.py --steps 2 --warmup 1 --ligands 12500000 --no-cpu-baseline --no-serial-leg > $OUT/b_shard12M_serial.json 2> $OUT/b_shard12M_serial.err
timeout 900 python bench.py --steps 1 --warmup 1 --pockets 16 --ligands 1253376 --no-cpu-baseline --no-serial-leg > $OUT/b_p16_shard.json 2> $OUT/b_p16_shard.err
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/b_1M.json 2> $OUT/b_1M.err
for f in $OUT/b_*.json; do python -c "import json; d=json.load(open('$f')); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],1))"; done
timeout 300 python tools/stress_shape.py 196 > $OUT/stress.log 2>&1; tail -n 1 $OUT/stress.log | cut -c1-50
PMX_BUDGET=512 timeout 300 python tools/stress_shape.py 196 > $OUT/stress_b512.log 2>&1; tail -n 1 $OUT/stress_b512.log | cut -c1-50
B="python bench.py --warmup 1 --steps 2 --no-cpu-baseline --no-serial-leg"
timeout 300 $B --conformers 64 --ligands 100352 > $OUT/b_6oim_c64.json 2> $OUT/b_6oim_c64.err
timeout 300 $B --conformers 16 --ligands 401408 > $OUT/b_6oim_c16.json 2> $OUT/b_6oim_c16.err
for f in $OUT/b_6oim*.json; do python -c "import json; d=json.load(open('$f')); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],1))"; done
