#!/bin/bash
OUT=gpurun_out/r4n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PMX_BENCH_DEVICE=0 PMX_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --ligands 500000 > $OUT/b_2ranks_gloo.json 2> $OUT/b_2ranks_gloo.err
tail -n 3 $OUT/b_2ranks_gloo.err; cat $OUT/b_2ranks_gloo.json | cut -c1-400
