#!/bin/bash
# round 6: mid-round check of the whole tree -- GPU suite (skip reasons listed), smoke, bench in both shapes, contact refusal rates
set -u
TAG=${1:-r6i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -8 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
timeout 300 python tools/contact_drop_rates.py > $OUT/contact_drop_rates.txt 2>&1; cat $OUT/contact_drop_rates.txt | cut -c1-260
python - <<PY
import json
for f in ("bench.json", "bench_driver_shape.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], [(k, d[k]["value"]) for k in ("extra", "extra2", "extra3") if k in d])
    if "cpu_baseline" in d: print(json.dumps(d["cpu_baseline"])[:1500])
PY
