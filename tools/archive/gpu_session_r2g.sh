#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 200 python tools/selfcol_ab.py > $OUT/selfcol_ab.txt 2>&1; cat $OUT/selfcol_ab.txt
bash tools/profile_r2.sh r2 > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
