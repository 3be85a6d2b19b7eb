#!/bin/bash
# round 4: AnymalTerrain observation columns by the scan kernel's threads (option fused_post) -- its GPU tests, the stand-in test on HIP, A/B in one session
out=gpurun_out/r4any; mkdir -p $out
python -m pytest tests -q -m gpu -k "anymal or Anymal" -x > $out/pytest_anymal.log 2>&1
echo "pytest anymal rc=$?"; tail -3 $out/pytest_anymal.log
MI_REFERENCE_ROOT=ab/ref_stage python -m pytest tests/test_gymapi_shim.py -q -x -k "anymal" > $out/pytest_shim.log 2>&1
echo "pytest shim rc=$?"; tail -3 $out/pytest_shim.log
for rep in 1 2; do
  for on in 1 0; do
    echo "== fused_post=$on rep$rep"
    MI_OPTS=fused_post=$on python tools/step_time.py AnymalTerrain:4096 AnymalTerrain:1024 AnymalTerrain:16384 2>/dev/null
  done
done > $out/anymal_obs_columns_ab.txt
cat $out/anymal_obs_columns_ab.txt
du -sh gpurun_out
