#!/bin/bash
# Time several builds of the library back to back on the same GPU box (clocks differ between boxes/sessions).
# Usage (on the GPU box): tools/ab_bench.sh ab/libA.so ab/libB.so ...
for rep in 1 2; do
  for lib in "$@"; do
    MI_ENGINE_LIB=$PWD/$lib python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$lib', 'rep$rep', 'Ant ms/step %.4f kernel %.4f | Humanoid ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['extra']['ms_per_step'], d['extra']['roofline']['kernel_ms']))"
  done
done
