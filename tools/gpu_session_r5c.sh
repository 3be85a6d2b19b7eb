#!/bin/bash
# round 5, third session: the whole GPU suite on the tree with the hand pairs / per-body hand masses / thumb orientation (as the driver runs it: -m gpu),
# smoke(), the per-body mass A/B, the host-wait A/B of the driver-shaped bench (HSA_ENABLE_INTERRUPT), the bench in both shapes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
for rep in 1 2 3; do
  for irq in 1 0; do
    HSA_ENABLE_INTERRUPT=$irq timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/b20_irq${irq}_$rep.json 2> $OUT/b20_irq${irq}_$rep.err
    python - <<PY
import json
d = json.loads(open("$OUT/b20_irq${irq}_$rep.json").read().strip().splitlines()[-1])
print("HSA_ENABLE_INTERRUPT=$irq rep $rep: value %.2f M  ms/step %.5f  median region %.5f  regions %s" % (d["value"] / 1e6, d["ms_per_step"], d["median_region"]["ms_per_step"], d["regions_ms_per_step"][:6]))
PY
  done
done 2>&1 | tee $OUT/host_wait_ab.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench_driver_shape.json", "bench.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e6, 2), d["ms_per_step"], d["pooled"]["ms_per_step"], [round(d[k]["value"] / 1e6, 2) for k in ("extra", "extra2", "extra3")], d["roofline"]["kernel_ms"])
PY
