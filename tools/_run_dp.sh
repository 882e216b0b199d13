set -u
O=gpurun_out/dp1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q --timeout 600 -k "golden or fresh or other_conformer or mixed or cut" > $O/tests.log 2>&1; echo "tests exit $?" >> $O/tests.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serial-leg > $O/bench.json 2> $O/bench.err
PMX_TREE_FLAGS=32768 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serial-leg > $O/bench_nodp.json 2> $O/bench_nodp.err
timeout 300 python bench.py --pockets 16 --ligands 200000 --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg > $O/p16.json 2> $O/p16.err
PMX_TREE_FLAGS=32768 timeout 300 python bench.py --pockets 16 --ligands 200000 --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg > $O/p16_nodp.json 2> $O/p16_nodp.err
tail -3 $O/tests.log
