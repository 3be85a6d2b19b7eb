#!/bin/bash
# round-2 session v: Ant / Humanoid step with the actor_params tensors switched off (default) against the library from before the
# joint-limit shifts (which read actor_scale unconditionally), then the GPU tests
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
one() { python bench.py --task $1 --num-envs $2 --steps $3 --warmup 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$4 $1@$2 %.4f ms/step  pooled %.4f' % (d['ms_per_step'], d['pooled']['ms_per_step']))"; }
for rep in 1 2 3; do
  for l in ab/lib_before_limshift.so isaacgymenvs_amd/libmi_engine.so; do
    MI_ENGINE_LIB=$PWD/$l one Ant 4096 2000 $(basename $l)
    MI_ENGINE_LIB=$PWD/$l one Humanoid 8192 500 $(basename $l)
  done
done | tee $OUT/actor_tensors_ab.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
