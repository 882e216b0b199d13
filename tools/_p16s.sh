run() { python bench.py --pockets 16 --ligands 200000 --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p16', d['value'])"; }
for s in 0 2048 8192 32768 1000000000; do echo COST=$s; PMX_BOUND_COST=$s run; done
