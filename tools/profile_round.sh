#!/bin/bash
# Run on the GPU box (through gpurun): kernel stats, the PMC passes (separate runs, --pmc only) and the bench line of the
# production pipeline, then tools/collect_profiles.py turns them into the tracked summaries under profiles/.
#   gpurun -- 'bash tools/profile_round.sh'
# then, in the repository (gpurun merges gpurun_out/ back, not profiles/):
#   O=gpurun_out/prof_r2; python tools/collect_profiles.py $O/bench.json $O/stats/*kernel_stats.csv $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ 200704
#   cp $O/fused_engine_traffic.json profiles/r2_fused_engine_traffic.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  PMX_PIPELINES=1 PMX_OVERLAP=0 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg > $OUT/pmc_$c.log 2>&1
done
PMX_PIPELINES=1 PMX_OVERLAP=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_SQ -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg > $OUT/pmc_SQ.log 2>&1
# the LDS-resident fused matcher (PMX_ENGINE=2): its HBM-side traffic, for DESIGN.md section 9
for c in FETCH_SIZE WRITE_SIZE; do
  PMX_ENGINE=2 rocprofv3 --pmc $c --output-format csv -d $OUT/fused_$c -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg > $OUT/fused_$c.log 2>&1
done
PMX_ENGINE=2 python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/fused_bench.json 2> $OUT/fused_bench.err
cd $ROOT
python - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/fused_%s/*counter_collection.csv" % c)[0]
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pmx::" in r["Kernel_Name"] and r["Counter_Name"] == c:
            tot[r["Kernel_Name"].split("(")[0].replace("void ", "")] += float(r["Counter_Value"])
    out[c] = tot
res = {"source": "PMX_ENGINE=2 (fused LDS-resident matcher), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --ligands 200000 --steps 1 --warmup 0",
       "ligands": 200704, "kernels": {}}
for k in out["FETCH_SIZE"]:
    f, w = out["FETCH_SIZE"][k], out["WRITE_SIZE"].get(k, 0.0)
    res["kernels"][k] = {"fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_ligand": (2 * f + w) * 1024 / 200704}
res["total_hbm_bytes_per_ligand"] = sum(v["hbm_bytes_per_ligand"] for v in res["kernels"].values())
try:
    res["bench"] = json.loads(open("$OUT/fused_bench.json").read().strip().splitlines()[-1])
    res["bench"] = {k: res["bench"][k] for k in ("value", "ms_per_step", "steps")}
except Exception as e:
    res["bench"] = repr(e)
json.dump(res, open("profiles/r2_fused_engine_traffic.json", "w"), indent=1)
json.dump(res, open("$OUT/fused_engine_traffic.json", "w"), indent=1)  # gpurun brings back gpurun_out/ only: copy it into profiles/ afterwards
print("fused engine:", res["total_hbm_bytes_per_ligand"], "B/ligand", res["bench"])
PY
STATS=$(ls $OUT/stats/*kernel_stats.csv | head -1)
python tools/collect_profiles.py $OUT/bench.json $STATS $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ 200704
tail -1 $OUT/bench.json | cut -c1-400; rm -f $OUT/pipelines.txt
# concurrency sweep for DESIGN.md section 3 (stream discipline)
for p in 1 2 3 4; do
  echo "pipelines=$p $(PMX_PIPELINES=$p python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $OUT/pipelines.txt
done
cat $OUT/pipelines.txt
