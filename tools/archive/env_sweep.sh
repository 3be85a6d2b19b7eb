#!/bin/bash
# Throughput against the env count on one GPU (run on the GPU box): the BASELINE sizes occupy 64-256 of the chip's 1024 SIMDs,
# this shows where the one-env-per-lane design saturates.  Usage: tools/env_sweep.sh > gpurun_out/env_sweep.txt
for spec in "Ant 1024 4096 16384 65536 262144" "Humanoid 2048 8192 32768 131072" "AnymalTerrain 4096 16384 65536" "ShadowHand 4096 16384 65536 131072"; do
  set -- $spec; task=$1; shift
  for n in "$@"; do
    steps=$(( 400000000 / n / 50 )); [ $steps -gt 1000 ] && steps=1000; [ $steps -lt 30 ] && steps=30
    timeout 300 python bench.py --task $task --num-envs $n --steps $steps --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = sys.stdin.readline()
if not l.strip(): print('$task', $n, 'FAILED'); sys.exit(0)
d = json.loads(l)
print('%-14s %8d envs  %8.4f ms/step  %8.1f M env-steps/s  (%d steps)' % ('$task', $n, d['ms_per_step'], d['value'] / 1e6, d['steps']))"
  done
done
