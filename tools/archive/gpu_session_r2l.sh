#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 200 python tools/hand_residency_ab.py 2>&1 | grep ShadowHand | tee $OUT/hand_time.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "hand or Hand" > $OUT/pytest_hand.log 2>&1; echo "hand rc=$?"; tail -15 $OUT/pytest_hand.log
