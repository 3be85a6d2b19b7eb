#!/bin/bash
# round 4, closing session: GPU suite with the reference files staged (stand-in, run-time assets, Articulation incl. the Franka arm), smoke(), the bench line
# (1000 steps and the driver's 20-step shape), the rocprofv3 trace + PMC passes summarised into profiles/, what kind of box this was.
set -u
TAG=${1:-r4z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{ echo "== rocminfo (GPU agent)"; rocminfo 2>/dev/null | grep -E "Compute Unit|Max Clock|Shader Engines|L2:|L3:|Max Waves Per CU" | tail -8
  echo "== rocm-smi idle"; rocm-smi --showclocks --showpower --showperflevel --showcomputepartition --showmemorypartition 2>&1 | grep -E "clock level|Power|Partition|Performance"; } > $OUT/box_info.txt 2>&1
( timeout 200 python tools/step_time.py ShadowHand:16384:20000 > $OUT/load.log 2>&1 & )
for i in $(seq 1 60); do sleep 3; grep -q "rep0" $OUT/load.log 2>/dev/null && break; done
{ echo "== rocm-smi under load (ShadowHand@16384 stepping: rep0 done, rep1 running)"; rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "clock level|Power|GPU use"; } >> $OUT/box_info.txt 2>&1
for i in $(seq 1 40); do sleep 3; grep -q "rep2" $OUT/load.log 2>/dev/null && break; done
{ echo "== step time of the load"; grep rep $OUT/load.log; } >> $OUT/box_info.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python -m pytest tests/test_gymapi_shim.py tests/test_runtime_assets.py tests/test_articulation.py -x -q > $OUT/pytest_shim.log 2>&1; echo "pytest shim (HIP backend) rc=$?"; tail -3 $OUT/pytest_shim.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
bash tools/profile_r4.sh $TAG > $OUT/profile.log 2>&1
python tools/summarize_profile.py $TAG > $OUT/summary.log 2>&1; tail -12 $OUT/summary.log
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.md profiles/traffic.json $OUT/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG
cat $OUT/box_info.txt
python - <<PY
import json
for f in ("bench.json", "bench_driver_shape.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("fp32", {}).get("frac"), [d[k]["value"] for k in ("extra", "extra2", "extra3")], d["box"])
PY
du -sh gpurun_out | tail -1
