run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import sys,json;b=json.loads(sys.stdin.read());k=b['roofline']['kernel_ms_per_launch'];print('$*', round(b['value']), round(b['ms_per_step'],1), [round(v,1) for v in k.values()])"; }
run PMX_BUDGET=1024
run PMX_BUDGET=256
run PMX_BUDGET=512
run PMX_BUDGET=4096
run PMX_MIN_LEVELS=2
run PMX_MIN_LEVELS=3
run PMX_SHARE_LEVELS=0
run PMX_SHARE_LEVELS=2
run PMX_CHUNK=131072
run PMX_CHUNK=524288
run PMX_OVERLAP=0
