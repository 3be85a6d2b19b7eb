#!/bin/bash
# round 6: one iteration on the Humanoid's limb-wave kernels -- its GPU tests, step time at 8192 envs, per-role phase times (instrumented library)
TAG=${1:-r6h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_multi_wave.py tests/test_gpu_fullsize.py tests/test_friction.py tests/test_gpu_parity.py tests/test_gpu_longrun.py -m gpu -q -k "umanoid or friction" > $OUT/pytest_humanoid.log 2>&1; tail -3 $OUT/pytest_humanoid.log
for r in 1 2; do timeout 300 python tools/step_time.py Humanoid:8192:1000 2>&1 | grep rep; done > $OUT/humanoid_step_time.txt; cat $OUT/humanoid_step_time.txt
if [ -f ab/lib_timing_mwc.so ]; then MI_ENGINE_LIB=$PWD/ab/lib_timing_mwc.so timeout 600 python tools/debug/mwc_phases.py > $OUT/humanoid_mwc_phases.txt 2>&1; head -24 $OUT/humanoid_mwc_phases.txt; fi
