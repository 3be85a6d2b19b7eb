set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_allegro_hand.py -x -q -k "hand or Hand" > gpurun_out/r4p/pytest_hand.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4p/pytest_hand.log
for rep in 1 2; do
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_r4_pre_select.so; do
    echo "== $lib rep$rep" >> gpurun_out/r4p/hand_pre_limits_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 300 python tools/step_time.py ShadowHand:16384:1000 AllegroHand:16384:800 2>&1 | grep "rep" >> gpurun_out/r4p/hand_pre_limits_ab.txt
  done
done
cat gpurun_out/r4p/hand_pre_limits_ab.txt
