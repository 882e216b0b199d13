#!/bin/bash
OUT=gpurun_out/r4k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/pack_scaling.py 64 > $OUT/pack_scaling.log 2>&1; cat $OUT/pack_scaling.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
python -c "import json; d=json.load(open('$OUT/b.json')); print('bench', round(d['value']/1e6,3), round(d['ms_per_step'],1)); print(d['end_to_end']); print(d['issue']); print(d['roofline']['traffic'])"
