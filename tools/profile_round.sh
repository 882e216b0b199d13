#!/bin/bash
# Run on the GPU box (through gpurun): the bench lines, kernel stats, the PMC passes (separate runs, --pmc only) of the
# round's build; tools/collect_profiles.py then turns them into the tracked summaries under profiles/.
#   gpurun -- 'bash tools/profile_round.sh'
# then, in the repository (gpurun merges gpurun_out/ back, not profiles/):
#   O=gpurun_out/prof_r6; PMX_PROFILE_TAG=r6 python tools/collect_profiles.py $O/bench.json $O/stats/*kernel_stats.csv $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ 200704
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.csrc_digest())" > $OUT/csrc_sha16.txt
python $ROOT/bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-leg --no-survey-leg --no-parity-sample > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_SQ -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/pmc_SQ.log 2>&1
# the shader clock under this load: GRBM_GUI_ACTIVE (summed over the XCDs) over the kernels' time (collect_profiles.py: "clock" of the SQ summary)
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_GRBM -o p -- python $ROOT/bench.py --ligands 200000 --steps 1 --warmup 0 --no-cpu-baseline --no-serial-leg --no-parity-sample > $OUT/pmc_GRBM.log 2>&1
# the device packer alone (graph builder + record writer) on 10^6 resident molecules of the bench generator's kind
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pack -o s -- python $ROOT/tools/pack_device_bench.py --molecules 1000000 --check > $OUT/pack.log 2>&1 </dev/null
# BASELINE configs[3] (16 pockets, per-GPU shard) and configs[4] (stress model, 64 conformers): the driver-reproducible lines
# SURVEY.md 8d-2's own library (tools/survey_library.py): the second headline line
python $ROOT/bench.py --library survey --steps 3 --warmup 1 > $OUT/bench_survey1M.json 2> $OUT/bench_survey1M.err
python $ROOT/bench.py --model stress64 --steps 3 --warmup 1 > $OUT/bench_stress64.json 2> $OUT/bench_stress64.err
python $ROOT/bench.py --pockets 16 --ligands 200000 --steps 1 --warmup 1 --no-serial-leg > $OUT/bench_pockets16.json 2> $OUT/bench_pockets16.err
ls -R $OUT | head -40
python $ROOT/bench.py --ligands 12500000 --steps 2 --warmup 1 --no-cpu-baseline --no-serial-leg > $OUT/bench_shard12M.json 2> $OUT/bench_shard12M.err
python $ROOT/bench.py --pockets 16 --ligands 1253376 --steps 1 --warmup 1 --no-serial-leg > $OUT/bench_pockets16_shard.json 2> $OUT/bench_pockets16_shard.err
