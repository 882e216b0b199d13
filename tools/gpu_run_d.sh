#!/bin/bash
set -x
OUT=gpurun_out/r4d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_variants.py::test_arena_pass_is_retried_with_the_arena_empty tests/test_gpu_variants.py::test_small_arena_changes_nothing_but_time -m gpu -q -x > $OUT/api.log 2>&1; echo "rc=$?" >> $OUT/api.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg"
PMX_OVERLAP=0 PMX_SUPER=1048576 timeout 300 $B > $OUT/b_one_chunk.json 2> $OUT/b_one_chunk.err
PMX_OVERLAP=0 timeout 300 $B > $OUT/b_no_overlap.json 2> $OUT/b_no_overlap.err
timeout 300 $B > $OUT/b_overlap.json 2> $OUT/b_overlap.err
PMX_LIG_SHARE=0.4 timeout 300 $B > $OUT/b_share40.json 2> $OUT/b_share40.err
PMX_LIG_SHARE=0.6 timeout 300 $B > $OUT/b_share60.json 2> $OUT/b_share60.err
PMX_SUPER=65536 timeout 300 $B > $OUT/b_super64k.json 2> $OUT/b_super64k.err
PMX_SUPER=262144 timeout 300 $B > $OUT/b_super256k.json 2> $OUT/b_super256k.err
tail -n 3 $OUT/api.log
for f in $OUT/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"; done
