#!/bin/bash
# round 4, last check of the tree as committed: the GPU tests touched after the closing session + the bench line in the driver's shape
out=gpurun_out/r4last; mkdir -p $out
python -m pytest tests/test_gpu_multi_wave.py -q -m gpu -k "anymal_terrain_observation" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
python bench.py --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$out/bench_driver_shape.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [d[k]["value"] for k in ("extra", "extra2", "extra3")], d["extra2"]["roofline"]["traffic"], d["extra3"]["roofline"]["traffic"])
PY
