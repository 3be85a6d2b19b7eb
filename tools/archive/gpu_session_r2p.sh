#!/bin/bash
# round-2 closing session: GPU tests, bench, Humanoid one- vs two-wave A/B, hand kernel A/B (round-1 kernel vs final), rocprof passes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('Ant', d['value']/1e6, d['ms_per_step'], 'pooled', d['pooled']['ms_per_step'])
for k in ('extra','extra2','extra3'):
    print(d[k]['workload'][:40], d[k]['value']/1e6, d[k]['ms_per_step'])
"
timeout 300 python tools/selfcol_ab.py 2>&1 | grep rep | tee $OUT/selfcol_ab.txt
for l in A D; do MI_ENGINE_LIB=$PWD/ab/lib_hand_$l.so timeout 200 python tools/hand_residency_ab.py 2>&1 | grep ShadowHand; done | tee $OUT/hand_ab.txt
bash tools/profile_r2.sh r2 > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
