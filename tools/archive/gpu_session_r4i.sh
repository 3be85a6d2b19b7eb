#!/bin/bash
# round 4, session i: the Ant's whole control step in ONE launch with post_physics_step spread over the four role waves (fused_sub + fused_post):
# bit-identity tests, same-session A/B through the engine option at several env counts.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py -x -q -k "fused" > $OUT/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_fused.log
for rep in 1 2; do
  for opt in 1 0; do
    echo "== fused_post=$opt rep$rep" >> $OUT/ant_fused_post_ab.txt
    MI_OPTS=fused_post=$opt timeout 300 python tools/step_time.py Ant:1024:3000 Ant:4096:3000 Ant:8192:2000 Ant:16384:1500 2>&1 | grep "rep" >> $OUT/ant_fused_post_ab.txt
  done
done
cat $OUT/ant_fused_post_ab.txt
du -sh gpurun_out | tail -1
