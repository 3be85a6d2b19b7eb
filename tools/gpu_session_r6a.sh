#!/bin/bash
# round 6, first session: the friction tangent-order fix in every engine form -- GPU suite, smoke, the bench line in both shapes
set -u
TAG=${1:-r6a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
python - <<PY
import json
for f in ("bench.json", "bench_driver_shape.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], [(k, d[k]["value"]) for k in ("extra", "extra2", "extra3") if k in d])
PY
