#!/bin/bash
# round 6: the scene kernel -- envs per workgroup as a launch-time choice (MI_SCENE_LANES A/B), broad phase, the boxes' half sizes in the LDS work area
set -u
TAG=${1:-r6s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_scene.py -m gpu -q > $OUT/pytest_scene.log 2>&1; echo "pytest scene rc=$?"; tail -3 $OUT/pytest_scene.log
{
  echo "== default choice of envs per workgroup (by batch size)"; timeout 900 python tools/scene_time.py 1024 4096 16384 2>&1 | grep FrankaCubeStack | cut -c1-330
  for L in 8 4 2 1; do echo "== MI_SCENE_LANES=$L"; MI_SCENE_LANES=$L timeout 900 python tools/scene_time.py 1024 4096 16384 2>&1 | grep FrankaCubeStack | cut -c1-330; done
  echo "== solver iterations 1+0 (default lanes): the sub-step without its sweeps"; MI_SCENE_ITERS=1,0 timeout 600 python tools/scene_time.py 4096 2>&1 | grep FrankaCubeStack | cut -c1-330
} > $OUT/scene_lanes_ab.txt 2>&1
cat $OUT/scene_lanes_ab.txt
