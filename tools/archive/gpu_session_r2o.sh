#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/selfcol_ab.py 2>&1 | grep rep | tee $OUT/selfcol_ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "Humanoid or humanoid or self_col or selfcol or locomotion" > $OUT/pytest_hum.log 2>&1; echo "hum rc=$?"; tail -12 $OUT/pytest_hum.log
