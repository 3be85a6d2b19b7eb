set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_allegro_hand.py -q > gpurun_out/r3u/pytest_allegro.log 2>&1; tail -30 gpurun_out/r3u/pytest_allegro.log
