#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 200 python tools/hand_residency_ab.py 2>&1 | grep ShadowHand | tee $OUT/hand_time.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2m/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in d.get("extra",{}).items() if isinstance(d.get("extra"),dict) else []:
    print(k, v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in("value","ms_per_step","env_steps_per_s")})
PY
