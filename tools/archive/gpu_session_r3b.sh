#!/bin/bash
# round 3, session b: first run of the Humanoid limb-per-wave kernel (engine_mwc.hpp) -- parity against the oracle in the block
# order, then Humanoid@8192 with option multi_wave = 2 (round 2: main + helper wave) against 32 (limb waves) in one process
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py -m gpu -q -x -k "humanoid or Humanoid" > $OUT/pytest_mw.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_mw.log
timeout 300 python tools/mwc_ab.py 2>&1 | tee $OUT/mwc_ab.txt
