#!/bin/bash
# round 6: the scene kernel with dense actor rows (one sweep loop over the actor slots), wave-uniform exit of the warm-start search
set -u
TAG=${1:-r6t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
  echo "== default envs per workgroup (4 up to 1024 envs, else 8)"; timeout 900 python tools/scene_time.py 1024 4096 16384 2>&1 | grep FrankaCubeStack | cut -c1-330
  echo "== MI_SCENE_LANES=4"; MI_SCENE_LANES=4 timeout 900 python tools/scene_time.py 4096 16384 2>&1 | grep FrankaCubeStack | cut -c1-330
  echo "== solver iterations 4+1 / 1+0: the sub-step with fewer / without its sweeps"
  MI_SCENE_ITERS=4,1 timeout 600 python tools/scene_time.py 4096 2>&1 | grep FrankaCubeStack | cut -c1-330
  MI_SCENE_ITERS=1,0 timeout 600 python tools/scene_time.py 4096 2>&1 | grep FrankaCubeStack | cut -c1-330
} > $OUT/scene_time.txt 2>&1
cat $OUT/scene_time.txt
