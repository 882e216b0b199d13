#!/bin/bash
# Knob sweep on the bench workload: prints value, ms per pass and per-chunk kernel times for each setting.
run() { env "$@" PMX_OVERLAP=0 timeout 200 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import sys,json;b=json.loads(sys.stdin.read());k=b['roofline']['kernel_ms_per_launch'];print('$*', round(b['value']), round(b['ms_per_step'],1), [round(v,1) for v in k.values()], round(b['roofline']['subtree_tasks_per_ligand'],2))"; }
for s in "${@:-PMX_BUDGET=1024}"; do run $s; done
