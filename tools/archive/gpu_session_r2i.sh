#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -k "ingenuity or Ingenuity" > $OUT/pytest_ingenuity.log 2>&1; echo "ingenuity rc=$?"; tail -30 $OUT/pytest_ingenuity.log
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
