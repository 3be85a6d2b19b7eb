#!/bin/bash
# round 5, fourth session: the whole GPU suite as the driver runs it (-m gpu) on the tree with AnymalTerrain's lagging dof-state tensor, smoke(),
# the lag's cost, the bench in both shapes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -12 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 300 python tools/anymal_lag_ab.py 4096 400 > $OUT/anymal_lag_ab.txt 2>&1; echo "lag ab rc=$?"; cat $OUT/anymal_lag_ab.txt
# PPO sanity runs on the changed physics (thumb where the asset puts it + hand-to-hand pairs; AnymalTerrain's lagging dof tensor): still learnable?
timeout 300 python examples/train_ppo.py --task ShadowHand --num-envs 16384 --iters 1200 --horizon 8 --epochs 5 > $OUT/ppo_shadow_hand.log 2>&1; echo "ppo hand rc=$?"; tail -2 $OUT/ppo_shadow_hand.log
timeout 300 python tools/hand_band_leavers.py > $OUT/hand_band_leavers.txt 2>&1; echo "leavers rc=$?"; tail -6 $OUT/hand_band_leavers.txt | cut -c1-250
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench_driver_shape.json", "bench.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e6, 2), d["ms_per_step"], d["pooled"]["ms_per_step"], [round(d[k]["value"] / 1e6, 2) for k in ("extra", "extra2", "extra3")], d["roofline"]["kernel_ms"])
PY
