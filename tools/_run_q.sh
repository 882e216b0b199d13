set -u
O=$GRAFT_REPO_ROOT/gpurun_out/q2; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q --timeout 600 -k "golden or other_conformer or stress or fp16" > $O/tests.log 2>&1; echo "tests exit $?" >> $O/tests.log
tail -3 $O/tests.log
timeout 300 python tools/stress_shape.py 40 > $O/stress.log 2>&1
tail -1 $O/stress.log
timeout 300 python tools/stress_shape.py 196 > $O/stress196.log 2>&1
tail -1 $O/stress196.log
timeout 300 python bench.py --conformers 64 --ligands 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg > $O/b64.json 2> $O/b64.err
python -c "
import json
d=json.load(open('$O/b64.json')); w=d['work']; print('b64', round(d['value']/1e6,3), round(d['ms_per_step'],1), w['wave_time_share'])"
