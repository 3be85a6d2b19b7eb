#!/bin/bash
# a box of the slow kind leaves socclk asleep under load: can the process (copy-engine traffic) or root (rocm-smi performance level) wake it?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/socclk
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 400 python tools/debug/socclk_experiment.py 2>&1 | grep -v "amdgpu.ids\|GPU Pipeline" > $OUT/part1.txt
cat $OUT/part1.txt
if grep -q "fast kind" $OUT/part1.txt; then exit 0; fi
{ echo "== performance level auto"; cat $OUT/part1.txt
  echo "== rocm-smi --setperflevel high"; rocm-smi --setperflevel high 2>&1 | grep -v "^=\|^$" | head -5
  timeout 200 python tools/step_time.py ShadowHand:16384:6000 2>&1 | grep rep; rocm-smi --showclocks 2>&1 | grep -E "socclk|sclk"
  rocm-smi --setperflevel auto 2>&1 | grep -v "^=\|^$" | head -3; } > $OUT/socclk_experiment.txt 2>&1
cat $OUT/socclk_experiment.txt
