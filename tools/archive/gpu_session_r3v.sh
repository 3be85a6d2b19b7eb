set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_gymapi_shim.py -q -m gpu -k "allegro or shadow" > gpurun_out/r3v/pytest_shim_hands.log 2>&1; tail -40 gpurun_out/r3v/pytest_shim_hands.log
