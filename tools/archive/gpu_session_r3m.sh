set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gymapi_shim.py -x -q 2>&1 | grep -v Warning | tail -40 > gpurun_out/shim_tests.log
cat gpurun_out/shim_tests.log
