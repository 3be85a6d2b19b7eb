#!/bin/bash
# round 4, session c: the Articulation task on the HIP backend (engine vs oracle, a run-time compiled hopper, the reference's unmodified humanoid_amp.py
# from the staged tree), the stand-in tests on HIP, the corrected lanes-per-env micro-benchmark.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_articulation.py -x -q > $OUT/pytest_articulation.log 2>&1; echo "pytest articulation rc=$?"; tail -3 $OUT/pytest_articulation.log
timeout 600 python -m pytest tests/test_gymapi_shim.py tests/test_runtime_assets.py -x -q > $OUT/pytest_shim.log 2>&1; echo "pytest shim rc=$?"; tail -3 $OUT/pytest_shim.log
for rep in 1 2 3; do timeout 120 tools/lanes/pgs_lanes_bench >> $OUT/lanes_per_env.txt 2>&1; echo "lanes rc=$?"; done; cat $OUT/lanes_per_env.txt
timeout 200 python tools/articulation_time.py 4096 16384 2>&1 | tail -6 | tee $OUT/articulation_step_time.txt
du -sh gpurun_out | tail -1
