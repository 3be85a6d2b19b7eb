#!/bin/bash
# Time several builds of the library back to back on the same GPU box (clocks differ between boxes/sessions).
# Usage (on the GPU box): tools/ab_bench.sh ab/libA.so ab/libB.so ...
for rep in 1 2; do
  for lib in "$@"; do
    MI_ENGINE_LIB=$PWD/$lib python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$lib', 'rep$rep', 'Ant %.4f | Humanoid %.4f | Anymal %.4f | Hand %.4f  (ms/step)' % (d['ms_per_step'], d['extra']['ms_per_step'], d['extra2']['ms_per_step'], d['extra3']['ms_per_step']))"
  done
done
