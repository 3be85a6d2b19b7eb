set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_runtime_assets.py -x -q -m gpu 2>&1 | grep -v Warning | tail -15 > gpurun_out/runtime_assets_gpu.log
cat gpurun_out/runtime_assets_gpu.log
