#!/bin/bash
# Profiling recipe (rounds 2 to 5) (run on the GPU box through gpurun): kernel-trace stats, separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ
# counters -- never combined with tracing), and the same two byte counters on the known-byte-count calibration kernels.
# Usage: tools/profile_r5.sh <tag>   ->  gpurun_out/prof_<tag>/ ; then tools/summarize_profile.py <tag> here.
set -u
TAG=${1:-r5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# the library these counters belong to: bench.py reports roofline.traffic / valu / fp32 / live_lanes only when the library it loads has this hash
sha256sum $GRAFT_REPO_ROOT/isaacgymenvs_amd/libmi_engine.so | cut -d' ' -f1 > $OUT/lib_sha256.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-shard-legs"
CAL=$GRAFT_REPO_ROOT/tools/calib/calib_fetch
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o $TAG -- $BENCH > $OUT/pmc_sq.log 2>&1
# round 4: counted fp32 operations and live lanes (SURVEY 8d achieved_fp32_fraction), its own SQ pass (8 slots)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_flops -o $TAG -- $BENCH > $OUT/pmc_flops.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o $TAG -- $CAL > $OUT/cal_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o $TAG -- $CAL > $OUT/cal_write.log 2>&1
ls -R $OUT | head -60
