#!/bin/bash
# round 5, scenes, second session: HIP scene tests after the warm start / joint velocity limits / launch-shape change, and the cost of the task at three sizes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5t; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_scene.py -m gpu -q -x > $OUT/pytest_scene.log 2>&1; echo "scene rc=$?"; tail -3 $OUT/pytest_scene.log
timeout 600 python tools/scene_time.py 1024 4096 16384 > $OUT/scene_time.txt 2>&1; echo "scene_time rc=$?"; grep FrankaCubeStack $OUT/scene_time.txt
