#!/bin/bash
# round 4, session d: GPU tests with the effort-limited hand drives in, their cost (same-session A/B through the engine option), bench line.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
for rep in 1 2; do
  for opt in 1 0; do
    echo "== drive_force_limit=$opt rep$rep" >> $OUT/hand_drive_clamp_ab.txt
    MI_OPTS=drive_force_limit=$opt timeout 300 python tools/step_time.py ShadowHand:16384:600 AllegroHand:16384:600 2>&1 | grep "rep" >> $OUT/hand_drive_clamp_ab.txt
  done
done
cat $OUT/hand_drive_clamp_ab.txt
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | cut -c1-600
du -sh gpurun_out | tail -1
