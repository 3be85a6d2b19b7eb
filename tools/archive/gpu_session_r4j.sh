#!/bin/bash
# round 4, session j: Humanoid -- post_physics_step on the role waves of the step's last sub-step launch (fused_post): bit-identity, A/B; Ant tests again
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py -x -q -k "fused or role_waves" > $OUT/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_fused.log
for rep in 1 2; do
  for opt in 1 0; do
    echo "== fused_post=$opt rep$rep" >> $OUT/humanoid_fused_post_ab.txt
    MI_OPTS=fused_post=$opt timeout 300 python tools/step_time.py Humanoid:8192:1500 Humanoid:4096:1500 Humanoid:16384:800 2>&1 | grep "rep" >> $OUT/humanoid_fused_post_ab.txt
  done
done
cat $OUT/humanoid_fused_post_ab.txt
