#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
MI_ENGINE_LIB=$PWD/ab/lib_timing.so timeout 300 python tools/debug/phase_timing_live.py > $OUT/humanoid_phases.txt 2>&1; cat $OUT/humanoid_phases.txt | tail -4
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
