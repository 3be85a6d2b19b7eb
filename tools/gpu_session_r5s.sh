#!/bin/bash
# round 5, scenes: the scene tests on HIP (oracle parity, push / grasp / stack, the reference's unmodified franka_cube_stack.py), the Articulation tests
# (per-link Jacobians), then the whole GPU suite as the driver runs it, and how long one simulate() of the Franka scene takes at 4096 envs
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_scene.py -m gpu -q -x > $OUT/pytest_scene.log 2>&1; echo "scene rc=$?"; tail -4 $OUT/pytest_scene.log
timeout 600 python -m pytest tests/test_articulation.py -m gpu -q > $OUT/pytest_articulation.log 2>&1; echo "articulation rc=$?"; tail -3 $OUT/pytest_articulation.log
timeout 300 python tools/scene_time.py > $OUT/scene_time.txt 2>&1; echo "scene_time rc=$?"; cat $OUT/scene_time.txt | tail -8
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
