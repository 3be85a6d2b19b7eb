#!/bin/bash
# round 4, closing numbers again (the closing session landed on a slow box): bench line (1000 steps, the driver's 20-step shape), rocprofv3 trace + PMC
# passes summarised into profiles/, the box's socclk under load.
set -u
TAG=${1:-r4z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 200 python tools/step_time.py ShadowHand:16384:12000 > $OUT/load.log 2>&1 & )
for i in $(seq 1 60); do sleep 3; grep -q "rep0" $OUT/load.log 2>/dev/null && break; done
{ echo "== rocm-smi under load (ShadowHand@16384 stepping)"; rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "clock level|Power|GPU use"; } > $OUT/box_info.txt 2>&1
for i in $(seq 1 40); do sleep 3; grep -q "rep2" $OUT/load.log 2>/dev/null && break; done
{ echo "== step time of the load"; grep rep $OUT/load.log; } >> $OUT/box_info.txt
cat $OUT/box_info.txt
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
bash tools/profile_r4.sh $TAG > $OUT/profile.log 2>&1
python tools/summarize_profile.py $TAG > $OUT/summary.log 2>&1; tail -8 $OUT/summary.log
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.md profiles/traffic.json $OUT/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG
python - <<PY
import json
for f in ("bench.json", "bench_driver_shape.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("fp32", {}).get("frac"), [d[k]["value"] for k in ("extra", "extra2", "extra3")])
PY
