#!/bin/bash
# Round-2 session F: after the aborted profile session -- plain tests and bench first, then rocprofv3 on top, each logged separately.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r2f -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
tail -3 $OUT/trace.log | cut -c1-200
