#!/bin/bash
# round 5, second session: the hand tests on HIP with the asset's hand-to-hand pairs in the kernels (and the thumb base's orientation fixed), their cost
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multi_wave.py tests/test_gpu_allegro_hand.py tests/test_allegro_hand.py -m gpu -q -k "hand or Hand" > $OUT/pytest_hand.log 2>&1; echo "pytest hand rc=$?"; tail -15 $OUT/pytest_hand.log
timeout 600 python tools/hand_pairs_ab.py 16384 400 > $OUT/hand_pairs_ab.txt 2>&1; echo "ab rc=$?"; cat $OUT/hand_pairs_ab.txt
timeout 300 python tools/hand_pairs_ab.py 4096 400 >> $OUT/hand_pairs_ab.txt 2>&1; tail -7 $OUT/hand_pairs_ab.txt
timeout 300 python tools/contact_drop_rates.py > $OUT/contact_drop_rates.txt 2>&1; cat $OUT/contact_drop_rates.txt | cut -c1-300
