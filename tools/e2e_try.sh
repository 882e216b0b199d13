#!/bin/bash
# The end-to-end legs of bench.py under different environments: `bash tools/e2e_try.sh` (on the GPU box). One line per variant.
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --no-survey-leg --no-cpu-baseline --no-parity-sample --steps 2 --warmup 1 > gpurun_out/e2e_$label.json 2> gpurun_out/e2e_$label.log </dev/null
  python - "$label" <<PY
import json,sys
d=json.loads(open("gpurun_out/e2e_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
e=d["end_to_end"]
print(sys.argv[1], round(d["ms_per_step"],2), "host-packed", round(e.get("overlapped_ligand_conformers_per_s",0)/1e6,2), "device-packed", round(e.get("device_packed_ligand_conformers_per_s",0)/1e6,2), e.get("device_packed_s"), e.get("device_packed_chunks"), [round(x, 4) for x in e.get("device_packed_host_s_in_pack_adopt_score_calls", [])], e.get("error"))
PY
}
run a PMX_BENCH_E2E_SHARES=1,3,4
run b PMX_BENCH_E2E_SHARES=1,3,4
run c PMX_BENCH_E2E_SHARES=1,2,5
