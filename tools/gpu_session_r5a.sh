#!/bin/bash
# round 5, first session: the GPU suite as the driver runs it (`-m gpu`: now includes the stand-in tests on the HIP backend), smoke(), and the bench
# line in both shapes on the rebuilt library -- the round's baseline before the kernel work.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "not perturbed_ant and not new_kinematic_tree and not franka and not force_sensors and not dextreme" > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
grep -c "test_gymapi_shim" $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench_driver_shape.json", "bench.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e6, 2), d["ms_per_step"], d["pooled"]["ms_per_step"], [round(d[k]["value"] / 1e6, 2) for k in ("extra", "extra2", "extra3")], d["roofline"]["kernel_ms"], d["roofline"]["traffic_source"])
    print(" shard_legs", {k: (round(v["ms_per_step"], 4), round(v["expected_8gpu_strong_scaling_efficiency"], 3)) for k, v in d.get("shard_legs", {}).items()})
    print(" ref_jit", d.get("cpu_baseline", {}).get("reference_jit_fns"), d.get("cpu_baseline", {}).get("threads_4"))
    print(" box", d["box"])
PY
