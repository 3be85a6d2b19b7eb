#!/bin/bash
# round 4, session f: hand_pre_kernel with the action rows staged through LDS as whole cache lines (same-session A/B against the build before it),
# GPU suite, the Ant / Humanoid trajectory band analysis on the HIP kernels.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_r4_base_pre.so; do
    echo "== $lib rep$rep" >> $OUT/hand_pre_stage_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 300 python tools/step_time.py ShadowHand:16384:600 AllegroHand:16384:600 2>&1 | grep "rep" >> $OUT/hand_pre_stage_ab.txt
  done
done
cat $OUT/hand_pre_stage_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o pre -- python $GRAFT_REPO_ROOT/tools/step_time.py ShadowHand:16384:300 > $OUT/trace.log 2>&1
grep -h "hand_pre_kernel\|hand_post_kernel\|hand_substep" $OUT/trace/*kernel_stats.csv | cut -c1-200 | head -5
rm -rf $OUT/trace
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
{ timeout 600 python tools/loco_band_leavers.py Ant 4096 12 cuda:0; timeout 900 python tools/loco_band_leavers.py Humanoid 8192 12 cuda:0; } 2>&1 | grep -v "GPU Pipeline\|amdgpu.ids" > $OUT/loco_band_leavers_hip.txt; tail -30 $OUT/loco_band_leavers_hip.txt
du -sh gpurun_out | tail -1
