#!/bin/bash
# round 4, session b: GPU tests after the cumulative-extras tensors (ABI 2), bench line, same-session A/Bs (Humanoid limb waves without the warm-start prefetch;
# the finger waves with SLP packing = v_pk_* ops), the lanes-per-env micro-benchmark, the profile with the fp32 / live-lane counters.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
for rep in 1 2; do
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_r4_nopf.so; do
    echo "== $lib rep$rep" >> $OUT/humanoid_prefetch_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 200 python tools/step_time.py Humanoid:8192:600 2>&1 | grep "rep" >> $OUT/humanoid_prefetch_ab.txt
  done
  for lib in isaacgymenvs_amd/libmi_engine.so ab/lib_r4_slp.so; do
    echo "== $lib rep$rep" >> $OUT/hand_slp_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 200 python tools/step_time.py ShadowHand:16384:600 2>&1 | grep "rep" >> $OUT/hand_slp_ab.txt
  done
done
cat $OUT/humanoid_prefetch_ab.txt $OUT/hand_slp_ab.txt
timeout 120 tools/lanes/pgs_lanes_bench > $OUT/lanes_per_env.txt 2>&1; echo "lanes rc=$?"; cat $OUT/lanes_per_env.txt
bash tools/profile_r4.sh r4b > $OUT/profile.log 2>&1
python tools/summarize_profile.py r4b > $OUT/summary.log 2>&1; tail -14 $OUT/summary.log
cp profiles/r4b_kernel_stats.csv profiles/r4b_pmc_summary.md profiles/traffic.json $OUT/ 2>/dev/null
rm -rf gpurun_out/prof_r4b
du -sh gpurun_out | tail -1
