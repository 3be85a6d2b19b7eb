#!/bin/bash
# round 4: fingertip states by the post kernel's own groups (option tips_in_post) -- hand GPU tests, the stand-in tests on HIP, A/B in one session
out=gpurun_out/r4tips; mkdir -p $out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_allegro_hand.py tests/test_gpu_parity.py tests/test_gpu_step_time_sanity.py -q -m gpu -k "hand or Hand" -x > $out/pytest_hand.log 2>&1
echo "pytest hand rc=$?"; tail -3 $out/pytest_hand.log
MI_REFERENCE_ROOT=ab/ref_stage python -m pytest tests/test_gymapi_shim.py -q -x -k "domain_randomisation or shadow_hand or allegro" > $out/pytest_shim.log 2>&1
echo "pytest shim rc=$?"; tail -3 $out/pytest_shim.log
for rep in 1 2; do
  for tip in 1 0; do
    echo "== tips_in_post=$tip rep$rep"
    MI_OPTS=tips_in_post=$tip python tools/step_time.py ShadowHand:16384 ShadowHand:4096 2>/dev/null
  done
done > $out/hand_tips_in_post_ab.txt
cat $out/hand_tips_in_post_ab.txt
du -sh gpurun_out
