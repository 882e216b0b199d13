run() { python bench.py --pockets 16 --ligands 200000 --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p16', d['value'])"; }
for s in 16384 32768 65536 131072; do echo SUPER=$s; PMX_SUPER=$s run; done
