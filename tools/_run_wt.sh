set -u
O=$GRAFT_REPO_ROOT/gpurun_out/wt; mkdir -p $O
cd $GRAFT_REPO_ROOT
PMX_CXXFLAGS="-DPMX_WALK_TICKS" python -m pharmaconet_amd.build --force > $O/build.log 2>&1
timeout 600 python tools/pocket_phases.py 100000 > $O/wt.log 2>&1
cut -c1-260 $O/wt.log | tail -18
