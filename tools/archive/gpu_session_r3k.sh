set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shadow_hand" 2>&1 | tail -5 > gpurun_out/hand_tests3.log
MI_ENGINE_LIB=$PWD/ab/lib_timing_hmw.so timeout 300 python tools/debug/hand_mw_phases.py > gpurun_out/hand_mw_phases3.txt 2>&1
timeout 600 python tools/hand_mw_ab.py 16384 > gpurun_out/hand_mw_ab4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3k -o r3k -- python $GRAFT_REPO_ROOT/bench.py --task ShadowHand --steps 300 --warmup 50 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r3k.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/hand_tests3.log gpurun_out/hand_mw_phases3.txt gpurun_out/hand_mw_ab4.txt
find gpurun_out/prof_r3k -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-160
