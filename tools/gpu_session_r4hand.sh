#!/bin/bash
# round 4: hand kernels (pre_physics_step on four lanes per env, option pre_parts) -- hand GPU tests, the stand-in hand tests on HIP, A/B in one session
out=gpurun_out/r4pre4; mkdir -p $out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_allegro_hand.py tests/test_gpu_parity.py tests/test_gpu_step_time_sanity.py -q -m gpu -k "hand or Hand" -x > $out/pytest_hand.log 2>&1
echo "pytest hand rc=$?"; tail -3 $out/pytest_hand.log
MI_REFERENCE_ROOT=ab/ref_stage python -m pytest tests/test_gymapi_shim.py -q -x -k "shadow_hand or allegro" > $out/pytest_shim.log 2>&1
echo "pytest shim rc=$?"; tail -3 $out/pytest_shim.log
for rep in 1 2; do
  for tip in 4 1; do
    echo "== pre_parts=$tip rep$rep"
    MI_OPTS=pre_parts=$tip python tools/step_time.py ShadowHand:16384 ShadowHand:4096 AllegroHand:16384 2>/dev/null
  done
done > $out/hand_pre4_ab.txt
cat $out/hand_pre4_ab.txt
du -sh gpurun_out
