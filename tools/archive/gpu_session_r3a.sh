#!/bin/bash
# round 3, session a: the limb-per-wave kernels with per-wave block sweeps (engine_mw.hpp P4) -- GPU parity against the oracle in
# the block order, then a same-session A/B of round 2's library (ab/lib_r2.so: one role sweeps all rows) against the tree's
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py tests/test_gpu_longrun.py "tests/test_gpu_parity.py" -m gpu -q -k "Ant or ant or Anymal or anymal or multi_wave or soak or benchmark_size or shard" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2; do
for lib in ab/lib_r2.so isaacgymenvs_amd/libmi_engine.so; do
  MI_ENGINE_LIB=$PWD/$lib timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>$OUT/bench_$(basename $lib).err | tee $OUT/bench_$(basename $lib)_$rep.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$lib', 'rep$rep', 'Ant %.4f (pooled %.4f, kern %.4f) | Humanoid %.4f | Anymal %.4f (kern %.4f) | Hand %.4f  (ms/step)' % (d['ms_per_step'], d['pooled']['ms_per_step'], d['roofline']['kernel_ms'], d['extra']['ms_per_step'], d['extra2']['ms_per_step'], d['extra2']['roofline']['kernel_ms'], d['extra3']['ms_per_step']))"
done
done 2>&1 | tee $OUT/ab.txt
