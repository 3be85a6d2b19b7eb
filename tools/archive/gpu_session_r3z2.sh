#!/bin/bash
# round 3, closing session z, second part: the PMC profile again with the byte-counter calibration probe built (tools/calib/calib_fetch was
# missing in session z: the summary fell back to factor 1.00 for FETCH_SIZE), then the bench line with the new profiles/traffic.json
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/profile_r3.sh r3z > $OUT/profile.log 2>&1
python tools/summarize_profile.py r3z > $OUT/summary.log 2>&1; tail -8 $OUT/summary.log
cp profiles/r3z_kernel_stats.csv profiles/r3z_pmc_summary.md profiles/traffic.json $OUT/
rm -rf gpurun_out/prof_r3z
timeout 600 python bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
