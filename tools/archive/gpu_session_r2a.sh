#!/bin/bash
# Round-2 session A (run on the GPU box through gpurun): GPU test-suite, bench, Humanoid self-collision on/off timing, kernel stats.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 200 python tools/selfcol_ab.py > $OUT/selfcol_ab.txt 2>&1; cat $OUT/selfcol_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r2a -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $OUT/trace.log 2>&1
ls $OUT/trace | head; find $OUT/trace -name "*kernel_stats*" | head -2 | xargs -I{} head -12 {}
