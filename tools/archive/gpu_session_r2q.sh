#!/bin/bash
# round-2 session q: hand sub-step A/B (HEAD / with actor_scale / with actor_scale + dof_limit_shift), then the GPU tests
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for l in ab/lib_hand_head.so ab/lib_hand_scales.so isaacgymenvs_amd/libmi_engine.so; do
    MI_ENGINE_LIB=$PWD/$l timeout 200 python tools/hand_residency_ab.py 2>&1 | grep ShadowHand
  done
done | tee $OUT/hand_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
