#!/bin/bash
# Round-2 session B: multi-wave sub-step tests + A/B against the single-wave kernel, self-collision timing after the broad phase.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export MI_MW_HAS16=1
timeout 600 python -m pytest tests/test_gpu_multi_wave.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 400 python tools/mw_ab.py > $OUT/mw_ab.txt 2>&1; cat $OUT/mw_ab.txt
timeout 200 python tools/selfcol_ab.py > $OUT/selfcol_ab.txt 2>&1; cat $OUT/selfcol_ab.txt
