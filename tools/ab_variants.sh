#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/ab_variants.sh name1 name2 ...'): tools/knob_sweep.py once per variants/libpmx_<name>.so
# (tools/build_variant.py), the product build first and last; one line per build in gpurun_out/ab/summary.txt.
# AB_SURVEY=1 adds a pass over SURVEY 8d-2's library (200 000 ligands) per build.
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/ab; mkdir -p $OUT; : > $OUT/summary.txt
export TMPDIR=/tmp
run() { # name, library ("" = the product build)
  if [ -n "$2" ]; then export PMX_LIBPMX=$2; else unset PMX_LIBPMX; fi
  timeout 300 python tools/knob_sweep.py --reps ${AB_REPS:-4} ${AB_ARGS:-} "PMX_TREE_FLAGS=16384" > $OUT/$1.log 2>&1
  line="$1: $(grep -E '^\(default\)' $OUT/$1.log | cut -c61-150) | tables: $(grep -E '^PMX_TREE_FLAGS' $OUT/$1.log | cut -c61-72)"
  if [ -n "${AB_SURVEY:-}" ]; then
    timeout 300 python tools/knob_sweep.py --library survey --ligands 200000 --reps 2 > $OUT/$1.survey.log 2>&1
    line="$line | survey200k: $(grep -E '^\(default\)' $OUT/$1.survey.log | cut -c61-150)"
  fi
  echo "$line" >> $OUT/summary.txt
}
run product ""
for n in "$@"; do run $n $PWD/variants/libpmx_$n.so; done
run product_again ""
cat $OUT/summary.txt
