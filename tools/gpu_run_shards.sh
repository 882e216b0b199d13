#!/bin/bash
OUT=gpurun_out/r4shards
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --ligands 12500000 --no-cpu-baseline --no-serial-leg > $OUT/b_shard12M.json 2> $OUT/b_shard12M.err
PMX_OVERLAP=0 timeout 900 python bench.py --steps 2 --warmup 1 --ligands 12500000 --no-cpu-baseline --no-serial-leg > $OUT/b_shard12M_serial.json 2> $OUT/b_shard12M_serial.err
timeout 1500 python bench.py --steps 1 --warmup 1 --pockets 16 --ligands 1253376 --no-cpu-baseline --no-serial-leg > $OUT/b_p16_shard.json 2> $OUT/b_p16_shard.err
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/b_1M.json 2> $OUT/b_1M.err
for f in $OUT/b_*.json; do python -c "import json; d=json.load(open('$f')); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],1))"; done
