set -u
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "1 1" "0 1" "0 0" "1 1"; do
  set -- $v
  PMX_CXXFLAGS="-DPMX_STAGE_CELL=$1 -DPMX_KEEP_CELL=$2" python -m pharmaconet_amd.build --force > $O/build.log 2>&1
  T=s$1k$2
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serial-leg > $O/bench_$T.json 2> $O/bench.err
  timeout 300 python bench.py --pockets 16 --ligands 100000 --steps 1 --warmup 1 --no-cpu-baseline --no-serial-leg > $O/p16_$T.json 2> $O/p16.err
  timeout 300 python tools/stress_shape.py 40 > $O/stress_$T.log 2>&1
  for f in bench p16; do python -c "
import json
d=json.load(open('$O/${f}_$T.json')); w=d['work']; print('$T $f', round(d['value']/1e6,3), round(d['ms_per_step'],1), w['wave_time_share'])"; done
  echo $T $(tail -1 $O/stress_$T.log | cut -c1-40)
done
