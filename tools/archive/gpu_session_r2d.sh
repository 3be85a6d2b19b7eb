#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== lanes16 kmax11"; MI_ENGINE_LIB=$PWD/ab/lib_l16k11.so timeout 200 python tools/selfcol_ab.py 2>&1 | grep rep1
  echo "== lanes32 kmax12"; MI_ENGINE_LIB=$PWD/ab/lib_lanes32.so timeout 200 python tools/selfcol_ab.py 2>&1 | grep rep1
done > $OUT/lanes_ab.txt 2>&1
cat $OUT/lanes_ab.txt
