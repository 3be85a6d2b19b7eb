#!/bin/bash
# round 3, session d: Humanoid limb waves, second version (dedicated pair wave, dense self-contact rows, one build stage) -- parity, A/B, phases
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py -m gpu -q -x -k "humanoid or Humanoid" > $OUT/pytest_mw.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_mw.log
timeout 300 python tools/mwc_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mwc_ab.txt
MI_ENGINE_LIB=$PWD/ab/lib_timing_mwc.so timeout 300 python tools/debug/mwc_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mwc_phases.txt
