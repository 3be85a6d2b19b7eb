#!/bin/bash
# final round-2 session: GPU tests, bench, hand A/B (old kernel vs new, same box), rocprof passes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 600 $OUT/bench.json; echo
for l in A C; do MI_ENGINE_LIB=$PWD/ab/lib_hand_$l.so timeout 200 python tools/hand_residency_ab.py 2>&1 | grep ShadowHand; done | tee $OUT/hand_ab.txt
bash tools/profile_r2.sh r2 > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
