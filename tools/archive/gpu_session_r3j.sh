set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shadow_hand" 2>&1 | tail -5 > gpurun_out/hand_tests2.log
MI_ENGINE_LIB=$PWD/ab/lib_timing_hmw.so timeout 300 python tools/debug/hand_mw_phases.py > gpurun_out/hand_mw_phases2.txt 2>&1
timeout 600 python tools/hand_mw_ab.py 16384 > gpurun_out/hand_mw_ab3.txt 2>&1
cat gpurun_out/hand_tests2.log gpurun_out/hand_mw_phases2.txt gpurun_out/hand_mw_ab3.txt
