#!/bin/bash
OUT=gpurun_out/r4z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_variants.py -m gpu -q -x --timeout 200 > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; grep "passed\|failed\|rc=" $OUT/tests.log
B="python bench.py --warmup 1 --steps 3 --no-cpu-baseline --no-serial-leg"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/b_$name.json 2> $OUT/b_$name.err; python -c "import json; d=json.load(open('$OUT/b_$name.json')); print('$name', round(d['value']/1e6,3), round(d['ms_per_step'],1), [round(x,1) for x in d['roofline']['kernel_ms_per_launch'].values()], round(d['work']['tree_frames_per_ligand'],1), round(d['work']['walker_passes_per_ligand'],1))"; }
run base X=1
grep "profiled pass" $OUT/b_base.err | sed 's/.*n_probes/n_probes/' | cut -c1-80
timeout 600 $B --steps 1 --pockets 16 --ligands 200000 > $OUT/p16.json 2> $OUT/p16.err
python -c "import json; d=json.load(open('$OUT/p16.json')); print('p16', round(d['value']/1e6,3), round(d['ms_per_step'],1))"
