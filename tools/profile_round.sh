#!/bin/bash
# Run on the GPU box (through gpurun): kernel stats, the PMC passes (separate runs, --pmc only) and the bench line of the
# production pipeline, then tools/collect_profiles.py turns them into the tracked summaries under profiles/.
#   gpurun -- 'bash tools/profile_round.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  PMX_PIPELINES=1 PMX_OVERLAP=0 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
PMX_PIPELINES=1 PMX_OVERLAP=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_SQ -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
cd $ROOT
STATS=$(ls $OUT/stats/*kernel_stats.csv | head -1)
python tools/collect_profiles.py $OUT/bench.json $STATS $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ 200704
tail -1 $OUT/bench.json | cut -c1-400
