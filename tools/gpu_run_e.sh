#!/bin/bash
set -x
OUT=gpurun_out/r4e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --no-cpu-baseline --no-serial-leg"
timeout 300 $B --steps 3 > $OUT/b_1m.json 2> $OUT/b_1m.err
timeout 300 $B --steps 2 --ligands 4000000 > $OUT/b_4m_overlap.json 2> $OUT/b_4m_overlap.err
PMX_OVERLAP=0 timeout 300 $B --steps 2 --ligands 4000000 > $OUT/b_4m_serial.json 2> $OUT/b_4m_serial.err
timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/b_p16_overlap.json 2> $OUT/b_p16_overlap.err
PMX_OVERLAP=0 timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/b_p16_serial.json 2> $OUT/b_p16_serial.err
PMX_LIG_SHARE=0.4 timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/b_p16_share40.json 2> $OUT/b_p16_share40.err
PMX_LIG_SHARE=0.6 timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/b_p16_share60.json 2> $OUT/b_p16_share60.err
PMX_CXXFLAGS=-DPMX_COUNTERS=2 python -m pharmaconet_amd.build --force > $OUT/build_c2.log 2>&1
timeout 300 $B --steps 1 > $OUT/b_c2.json 2> $OUT/b_c2.err
for f in $OUT/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value']/1e6, d['ms_per_step'])"; done
grep "profiled pass" $OUT/b_c2.err | sed 's/.*dbg/dbg/'
