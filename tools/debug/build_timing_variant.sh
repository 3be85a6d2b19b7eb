#!/bin/bash
# ab/lib_timing.so = the current objects with kernels_humanoid.hip rebuilt with -DMI_TIMING (s_memtime stamps per sub-step phase)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
B=$ROOT/isaacgymenvs_amd/csrc/build
cd $ROOT/isaacgymenvs_amd/csrc
FLAGS=$(python -c "import sys; sys.path.insert(0, '$ROOT'); from isaacgymenvs_amd import native; print(' '.join(native.HIPCC_FLAGS))" 2>/dev/null || echo "--offload-arch=gfx950 -O3 -std=c++17 -fPIC")
hipcc $FLAGS -DMI_TIMING -c kernels_humanoid.hip -o /tmp/hum_timing.o
mkdir -p $ROOT/ab
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v kernels_humanoid) /tmp/hum_timing.o -o $ROOT/ab/lib_timing.so
echo built $ROOT/ab/lib_timing.so
