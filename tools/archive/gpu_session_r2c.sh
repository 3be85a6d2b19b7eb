#!/bin/bash
# Round-2 session C: Humanoid with 16 vs 32 envs per wave (self-collision on / off), full GPU test-suite on the default build.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export MI_MW_HAS16=1
for rep in 1 2; do
  echo "== lanes16 (default build)"; timeout 200 python tools/selfcol_ab.py 2>&1 | grep rep1
  echo "== lanes32"; MI_ENGINE_LIB=$PWD/ab/lib_lanes32.so timeout 200 python tools/selfcol_ab.py 2>&1 | grep rep1
done > $OUT/lanes_ab.txt 2>&1
cat $OUT/lanes_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
