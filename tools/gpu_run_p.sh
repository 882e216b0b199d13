#!/bin/bash
OUT=gpurun_out/r4p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_variants.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; grep -v "^ \|^$" $OUT/tests.log | tail -n 12
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
timeout 300 $B > $OUT/b.json 2> $OUT/b.err
PMX_TREE_FLAGS=1024 timeout 300 $B > $OUT/b_nopath.json 2> $OUT/b_nopath.err
for f in b b_nopath; do python -c "import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],1), d['roofline']['kernel_ms_per_launch'].values(), d['work']['tree_frames_per_ligand'], d['work']['walker_passes_per_ligand'])"; done
grep "profiled pass" $OUT/b.err | sed 's/.*n_exact_values/n_exact_values/'
timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/p16.json 2> $OUT/p16.err
python -c "import json; d=json.load(open('$OUT/p16.json')); print('p16', round(d['value']/1e6,3), round(d['ms_per_step'],1))"
