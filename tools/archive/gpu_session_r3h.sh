set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shadow_hand" 2>&1 | tail -15 > gpurun_out/hand_tests.log
timeout 600 python tools/hand_mw_ab.py 16384 > gpurun_out/hand_mw_ab.txt 2>&1
cat gpurun_out/hand_tests.log; cat gpurun_out/hand_mw_ab.txt
