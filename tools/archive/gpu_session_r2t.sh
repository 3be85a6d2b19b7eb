#!/bin/bash
# round-2 session t: A/B of LDS-staged observation rows in loco_post_kernel (-DMI_POST_STAGE, ab/lib_post_stage.so), then GPU tests, bench,
# smoke, rocprof passes with the final tree
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2t
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
one() { python bench.py --task $1 --num-envs $2 --steps $3 --warmup 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$4 $1@$2 %.4f ms/step  pooled %.4f' % (d['ms_per_step'], d['pooled']['ms_per_step']))"; }
for rep in 1 2 3; do
  for l in isaacgymenvs_amd/libmi_engine.so ab/lib_post_stage.so; do
    MI_ENGINE_LIB=$PWD/$l one Ant 4096 2000 $(basename $l)
    MI_ENGINE_LIB=$PWD/$l one Humanoid 8192 500 $(basename $l)
  done
done | tee $OUT/post_stage_ab.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('Ant', d['value']/1e6, d['ms_per_step'], 'pooled', d['pooled']['ms_per_step'])
for k in ('extra','extra2','extra3'):
    print(d[k]['workload'][:40], d[k]['value']/1e6, d[k]['ms_per_step'])
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_r2.sh r2t > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
