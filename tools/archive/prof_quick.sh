#!/bin/bash
# rocprofv3 kernel-trace stats of tools/step_time.py runs (on the GPU box).  Usage: tools/prof_quick.sh <tag> Task:n[:steps] ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/tools/step_time.py "$@" > $R/gpurun_out/prof_$TAG/log.txt 2>&1
grep -v Forcing $R/gpurun_out/prof_$TAG/log.txt | grep "rep2"
cp $R/gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
grep "mi::" $R/gpurun_out/${TAG}_kernel_stats.csv | grep -v "init_\|reset_" | awk -F'","' '{printf "%-110s calls %6s avg %9.1f ns\n", substr($1,2,110), $2, $4}' | head -24
