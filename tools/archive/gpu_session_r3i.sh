set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MI_ENGINE_LIB=$PWD/ab/lib_timing_hmw.so timeout 300 python tools/debug/hand_mw_phases.py > gpurun_out/hand_mw_phases.txt 2>&1
timeout 600 python tools/hand_mw_ab.py 16384 > gpurun_out/hand_mw_ab2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3i -o r3i -- python $GRAFT_REPO_ROOT/bench.py --task ShadowHand --steps 300 --warmup 50 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r3i.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/hand_mw_phases.txt gpurun_out/hand_mw_ab2.txt; tail -3 gpurun_out/prof_r3i.log
find gpurun_out/prof_r3i -name "*kernel_stats.csv" | head -1 | xargs head -12
