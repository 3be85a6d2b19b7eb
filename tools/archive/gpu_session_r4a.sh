#!/bin/bash
# round 4, session a: the refactored task kernels (per-env bodies in csrc/tasks/*_step.hpp / hand_task.hpp), the stand-in tests on the HIP backend with the
# reference tree staged (tools/debug/stage_reference.sh), bench line with the corrected compute roofline, profile incl. the fp32 / live-lane counters,
# contact-slot drop rates per limb.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python -m pytest tests/test_gymapi_shim.py -q > $OUT/pytest_shim_hip.log 2>&1; echo "pytest shim rc=$?"; tail -3 $OUT/pytest_shim_hip.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err; echo "bench20 rc=$?"
# same-session A/B: the (env, column group) post kernel of the hand tasks against the one-lane-per-env form (ab/lib_r4_base.so = this tree before that change)
for rep in 1 2; do
  for lib in ab/lib_r4_base.so isaacgymenvs_amd/libmi_engine.so; do
    echo "== $lib rep$rep" >> $OUT/hand_post_ab.txt
    MI_ENGINE_LIB=$PWD/$lib timeout 200 python tools/step_time.py ShadowHand:16384:600 AllegroHand:16384:600 >> $OUT/hand_post_ab.txt 2>&1
  done
done
grep -v "^GPU Pipeline" $OUT/hand_post_ab.txt | tail -30
timeout 300 python tools/contact_drop_rates.py > $OUT/contact_drop_rates.txt 2>&1; echo "drops rc=$?"; tail -12 $OUT/contact_drop_rates.txt
bash tools/profile_r4.sh r4a > $OUT/profile.log 2>&1
python tools/summarize_profile.py r4a > $OUT/summary.log 2>&1; tail -12 $OUT/summary.log
cp profiles/r4a_kernel_stats.csv profiles/r4a_pmc_summary.md profiles/traffic.json $OUT/ 2>/dev/null
rm -rf gpurun_out/prof_r4a/*/*/*.db 2>/dev/null
du -sh gpurun_out | tail -1
