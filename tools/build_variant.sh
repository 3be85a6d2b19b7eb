#!/bin/bash
# Build a variant of libmi_engine.so from a source tree (for A/B timing inside one gpurun session).
# Usage: tools/build_variant.sh <repo-root-of-sources> <out.so> [extra hipcc flags...]
set -e
SRC=$1; OUT=$2; shift 2
B=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize $@"
for f in $(cd $SRC/isaacgymenvs_amd/csrc && ls *.hip | sed 's/\.hip$//'); do
  ( cd $SRC/isaacgymenvs_amd/csrc && hipcc $FLAGS -c $f.hip -o $B/$f.o 2>/dev/null ) &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $OUT
rm -rf $B
echo built $OUT
