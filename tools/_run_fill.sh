set -u
O=$GRAFT_REPO_ROOT/gpurun_out/fill; mkdir -p $O
cd $GRAFT_REPO_ROOT
PMX_CXXFLAGS=-DPMX_TABLE_FILL python -m pharmaconet_amd.build --force > $O/build.log 2>&1
PMX_RAW_DBG=1 timeout 600 python tools/pocket_phases.py 50000 > $O/fill.log 2>&1
tail -20 $O/fill.log
