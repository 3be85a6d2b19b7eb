#!/bin/bash
# round 4, session l: AnymalTerrain's curriculum pre-pass on the trunk wave of the fused launch (one kernel less): tests, A/B against the build before it
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "Anymal or anymal or terrain or sub_steps" > $OUT/pytest_anymal.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_anymal.log
for rep in 1 2; do
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_r4_pre_tail.so; do
    echo "== $lib rep$rep" >> $OUT/anymal_cmdnorm_tail_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 300 python tools/step_time.py AnymalTerrain:4096:1500 AnymalTerrain:1024:1500 2>&1 | grep "rep" >> $OUT/anymal_cmdnorm_tail_ab.txt
  done
done
cat $OUT/anymal_cmdnorm_tail_ab.txt
