#!/bin/bash
# ab/lib_timing_mwc.so = the current objects with kernels_humanoid_mwc.hip rebuilt with -DMI_TIMING (s_memtime stamps per role and phase)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
B=$ROOT/isaacgymenvs_amd/csrc/build
cd $ROOT/isaacgymenvs_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize"
hipcc $FLAGS -DMI_TIMING -c kernels_humanoid_mwc.hip -o /tmp/hum_mwc_timing.o 2>/dev/null
mkdir -p $ROOT/ab
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v "kernels_humanoid_mwc.o" | grep -v "/cpu_") /tmp/hum_mwc_timing.o -o $ROOT/ab/lib_timing_mwc.so
echo built $ROOT/ab/lib_timing_mwc.so
