#!/bin/bash
# Development helper: recompile only the named translation units of isaacgymenvs_amd/csrc (in parallel) and relink libmi_engine.so, instead of
# native.build()'s "any header changed -> everything" rule (6 minutes).  The caller knows which units a header change reaches.
# Usage: tools/dev_rebuild.sh kernels_ant kernels_humanoid ...
set -e
cd "$(dirname "$0")/../isaacgymenvs_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
pids=()
for tu in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -c $tu.hip -o build/$tu.o > build/$tu.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; tail -20 build/*.log | grep -B2 -A8 "error" | head -60; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "build/cpu_") -o ../libmi_engine.so.tmp
mv ../libmi_engine.so.tmp ../libmi_engine.so
echo "relinked $(ls -la ../libmi_engine.so)"
