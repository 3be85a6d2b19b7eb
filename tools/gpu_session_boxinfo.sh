#!/bin/bash
# what kind of box is this: rocm-smi clocks idle and under a known load (ShadowHand@16384 stepping), the step time of that load, the bench probe values
set -u
TAG=${1:-box}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{ echo "== rocm-smi idle"; rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -E "clock level|Power|Performance"; } > $OUT/box_info.txt 2>&1
( timeout 200 python tools/step_time.py ShadowHand:16384:20000 > $OUT/load.log 2>&1 & )
for i in $(seq 1 60); do sleep 3; grep -q "rep0" $OUT/load.log 2>/dev/null && break; done
{ echo "== rocm-smi under load (ShadowHand@16384 stepping)"; rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "clock level|Power|GPU use"
  echo "== rocm-smi -a excerpts under load"; rocm-smi --showclkfrq 2>&1 | grep -E "^GPU|\*" | head -40; rocm-smi --showvoltage --showtemp --showmaxpower 2>&1 | grep -E "^GPU" | head -20; } >> $OUT/box_info.txt 2>&1
for i in $(seq 1 40); do sleep 3; grep -q "rep2" $OUT/load.log 2>/dev/null && break; done
{ echo "== step time of the load"; grep rep $OUT/load.log; } >> $OUT/box_info.txt
cat $OUT/box_info.txt
