#!/bin/bash
# round 4, session m: the ShadowHand's fingertip states computed by the finger waves of the last sub-step launch (no hand_tips_kernel): tests, A/B
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_multi_wave.py tests/test_gymapi_shim.py -x -q -k "hand or Hand" > $OUT/pytest_hand.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_hand.log
for rep in 1 2; do
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_r4_pre_tips.so; do
    echo "== $lib rep$rep" >> $OUT/hand_tips_in_launch_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 300 python tools/step_time.py ShadowHand:16384:1000 ShadowHand:4096:1000 2>&1 | grep "rep" >> $OUT/hand_tips_in_launch_ab.txt
  done
done
cat $OUT/hand_tips_in_launch_ab.txt
