#!/bin/bash
# round 6: the Ant's one-launch kernel compiled with other machine-scheduler strategies (kernels_mw_ant.hip only; same arithmetic, other instruction order)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6sched; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for rep in 1 2 3; do
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_sched_ilp.so ab/lib_sched_clause.so ab/lib_sched_bias0.so; do
    echo "== $lib rep $rep"; MI_ENGINE_LIB=$PWD/$lib timeout 300 python tools/step_time.py Ant:4096:2000 2>&1 | grep -E "ms/step|ms per step" | tail -3
  done
done
} > $OUT/ant_sched_ab.txt 2>&1
cat $OUT/ant_sched_ab.txt
