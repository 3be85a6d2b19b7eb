#!/bin/bash
# round 6, validation session: the GPU suite as the driver runs it (-m gpu; the reference files staged: stand-in, run-time assets, Articulation), smoke(),
# the bench line (1000 steps and the driver's 20-step shape), the rocprofv3 trace + PMC passes summarised into profiles/ (stamped with the library's
# content hash), contact statistics, what kind of box this was.
set -u
TAG=${1:-r6g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{ echo "== rocminfo (GPU agent)"; rocminfo 2>/dev/null | grep -E "Compute Unit|Max Clock|Shader Engines|L2:|L3:|Max Waves Per CU" | tail -8
  echo "== rocm-smi idle"; rocm-smi --showclocks --showpower --showperflevel --showcomputepartition --showmemorypartition 2>&1 | grep -E "clock level|Power|Partition|Performance"; } > $OUT/box_info.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
grep -c "PASSED\|passed" $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
bash tools/profile_r5.sh $TAG > $OUT/profile.log 2>&1
python tools/summarize_profile.py $TAG > $OUT/summary.log 2>&1; tail -12 $OUT/summary.log
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.md profiles/traffic.json $OUT/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG
# the bench again, now that profiles/traffic.json carries this library's hash: the counter-derived fields are present
timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench_with_counters.json 2> $OUT/bench_with_counters.err; echo "bench (counters) rc=$?"
timeout 300 python tools/contact_drop_rates.py > $OUT/contact_drop_rates.txt 2>&1; cat $OUT/contact_drop_rates.txt | cut -c1-260
cat $OUT/box_info.txt
# the scene kernel: the reference's franka_cube_stack.py at 4096 envs, and how much of a sub-step its sweeps are (solver iteration counts 8+1 / 4+1 / 1+0)
{ timeout 600 python tools/scene_time.py 4096; MI_SCENE_ITERS=4,1 timeout 600 python tools/scene_time.py 4096; MI_SCENE_ITERS=1,0 timeout 600 python tools/scene_time.py 4096; } 2>&1 | grep FrankaCubeStack | cut -c1-330 > $OUT/scene_time.txt; cat $OUT/scene_time.txt
python - <<PY
import json
for f in ("bench.json", "bench_driver_shape.json", "bench_with_counters.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("fp32", {}).get("frac"), d["roofline"].get("traffic"), [d[k]["value"] for k in ("extra", "extra2", "extra3")], d["box"])
PY
du -sh gpurun_out | tail -1
# bench.py's N > 1 path with two ranks on this one GPU (gloo; RCCL refuses two ranks on one device): barriers, the MAX all-reduce of the region time, reducers
bash tools/debug/bench_two_ranks_one_gpu.sh > $OUT/bench_two_ranks_one_gpu.txt 2>&1; tail -6 $OUT/bench_two_ranks_one_gpu.txt | cut -c1-300
