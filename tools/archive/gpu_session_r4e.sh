#!/bin/bash
# round 4, session e: what kind of box this is (rocminfo / rocm-smi under load next to the probe values), the Articulation tests on HIP incl. the Franka
# arm from its URDF meshes, the ShadowHand band-leaver analysis at the benchmark size on the HIP kernels.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{ echo "== rocminfo (GPU agents)"; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Wavefront Size|Workgroup Max Size|Max Waves Per CU|Shader Engines|Shader Arrs|Cacheline|L2:|L3:|Size:.*KB" | head -60
  echo "== rocm-smi idle"; rocm-smi --showclocks --showpower --showperflevel --showcomputepartition --showmemorypartition 2>&1 | head -60; } > $OUT/box_info.txt 2>&1
( timeout 120 python tools/step_time.py ShadowHand:16384:3000 > $OUT/load.log 2>&1 & )
sleep 45
{ echo "== rocm-smi under load (ShadowHand@16384 stepping)"; rocm-smi --showclocks --showpower --showuse 2>&1 | head -60; } >> $OUT/box_info.txt 2>&1
sleep 40
{ echo "== step time of the load"; grep rep $OUT/load.log; } >> $OUT/box_info.txt
timeout 900 python -m pytest tests/test_articulation.py -x -q > $OUT/pytest_articulation.log 2>&1; echo "pytest articulation rc=$?"; tail -3 $OUT/pytest_articulation.log
timeout 900 python tools/hand_band_leavers.py 16384 24 cuda:0 2>&1 | grep -v "^GPU Pipeline" > $OUT/hand_band_leavers_hip.txt; echo "leavers rc=$?"; tail -12 $OUT/hand_band_leavers_hip.txt
cat $OUT/box_info.txt | tail -40
du -sh gpurun_out | tail -1
