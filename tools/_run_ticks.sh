set -u
O=$GRAFT_REPO_ROOT/gpurun_out/ticks; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/gpu_tests.log 2>&1; echo "suite rc=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
timeout 600 python tools/pocket_phases.py 100000 > $O/phases_plain.log 2>&1
PMX_CXXFLAGS=-DPMX_TABLE_TICKS python -m pharmaconet_amd.build --force > $O/build.log 2>&1
timeout 600 python tools/pocket_phases.py 100000 > $O/phases_ticks.log 2>&1
tail -20 $O/phases_ticks.log
