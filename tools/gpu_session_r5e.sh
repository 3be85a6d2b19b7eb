#!/bin/bash
# round 5: the Articulation tests on HIP with the kuka_allegro arm + hand (run-time robot, prebuilt variant), then the whole GPU suite once more
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_articulation.py -m gpu -q > $OUT/pytest_articulation.log 2>&1; echo "articulation rc=$?"; tail -4 $OUT/pytest_articulation.log
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
