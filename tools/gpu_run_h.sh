#!/bin/bash
set -x
OUT=gpurun_out/r4h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --warmup 1 --steps 2 --no-cpu-baseline --no-serial-leg"
timeout 600 $B --conformers 64 --ligands 100352 > $OUT/b_6oim_c64.json 2> $OUT/b_6oim_c64.err
timeout 600 $B --conformers 32 --ligands 200704 > $OUT/b_6oim_c32.json 2> $OUT/b_6oim_c32.err
timeout 600 $B --conformers 16 --ligands 401408 > $OUT/b_6oim_c16.json 2> $OUT/b_6oim_c16.err
timeout 600 python tools/stress_shape.py 196 > $OUT/stress_full.log 2>&1
for f in $OUT/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value']/1e6, d['ms_per_step'], d['work']['wave_time_share'], d['work']['tree_frames_per_ligand'], d['work']['walker_passes_per_ligand'])"; done
tail -n 1 $OUT/stress_full.log
