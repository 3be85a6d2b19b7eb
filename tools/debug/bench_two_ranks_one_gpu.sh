#!/bin/bash
# bench.py's N > 1 path with TWO ranks on ONE GPU (gloo instead of RCCL, both ranks LOCAL_RANK=0): barriers, MAX all-reduce of the region time, the settle
# loop's broadcast, reducers, side-config sharding, the non-zero rank's exit path.  What it cannot show: RCCL itself (one-rank group: tests/test_gpu_step_time_sanity.py).
cd $(dirname $0)/../..
export MI_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/bench_rank1.out 2> /tmp/bench_rank1.err &
RANK=0 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/bench_rank0.out 2> /tmp/bench_rank0.err
rc0=$?
wait
echo "rank0 rc=$rc0; rank1 stdout bytes: $(wc -c < /tmp/bench_rank1.out)"
tail -3 /tmp/bench_rank0.err; tail -3 /tmp/bench_rank1.err
python - <<PY
import json
lines = [l for l in open("/tmp/bench_rank0.out").read().splitlines() if l.strip()]
d = json.loads(lines[-1])
print("n_gpus", d["n_gpus"], "value", d["value"], "ms_per_step", d["ms_per_step"], "scaling", d["scaling"], "config", d["config"]["workload"][:80])
print("job_stats", d.get("job_stats"))
for k in ("extra", "extra2", "extra3"):
    print(k, d[k]["value"], d[k].get("job_stats"))
PY
