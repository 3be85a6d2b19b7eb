#!/bin/bash
# round-2 session r: hand sub-step A/B (HEAD before the actor_params tensors / scales only / scales + limit shifts), GPU tests, bench
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for l in ab/lib_hand_head.so ab/lib_hand_scales.so isaacgymenvs_amd/libmi_engine.so; do
    MI_ENGINE_LIB=$PWD/$l timeout 200 python tools/hand_residency_ab.py 2>&1 | grep ShadowHand
  done
done | tee $OUT/hand_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('Ant', d['value']/1e6, d['ms_per_step'], 'pooled', d['pooled']['ms_per_step'])
for k in ('extra','extra2','extra3'):
    print(d[k]['workload'][:40], d[k]['value']/1e6, d[k]['ms_per_step'])
"
