#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "ball_balance or BallBalance" > $OUT/pytest_bbot.log 2>&1; echo "bbot rc=$?"; tail -40 $OUT/pytest_bbot.log
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
