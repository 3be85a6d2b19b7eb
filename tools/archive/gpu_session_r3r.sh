set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multi_wave.py tests/test_gpu_parity.py -x -q -m gpu -k "ant or Ant" 2>&1 | tail -5 > gpurun_out/ant_fused_tests.log
timeout 600 python tools/ant_fused_post_ab.py 4096 1024 16384 > gpurun_out/ant_fused_post_ab.txt 2>&1
cat gpurun_out/ant_fused_tests.log gpurun_out/ant_fused_post_ab.txt
