#!/bin/bash
# round 3, closing session z (fused sub-steps, terrain walls): bench line + PMC profile, summarised ON THE BOX (the raw rocprofv3 traces exceed what gpurun copies back)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
bash tools/profile_r3.sh r3z > $OUT/profile.log 2>&1
python tools/summarize_profile.py r3z > $OUT/summary.log 2>&1; tail -12 $OUT/summary.log
cp profiles/r3z_kernel_stats.csv profiles/r3z_pmc_summary.md profiles/traffic.json $OUT/
rm -rf gpurun_out/prof_r3z
du -sh gpurun_out
