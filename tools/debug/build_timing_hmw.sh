#!/bin/bash
# ab/lib_timing_hmw.so = the current objects with kernels_shadow_hand_mw.hip rebuilt with -DMI_TIMING (s_memtime stamps per role and phase)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
B=$ROOT/isaacgymenvs_amd/csrc/build
cd $ROOT/isaacgymenvs_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize"
hipcc $FLAGS -DMI_TIMING -c kernels_shadow_hand_mw.hip -o /tmp/hand_mw_timing.o 2>/dev/null
mkdir -p $ROOT/ab
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v "kernels_shadow_hand_mw.o" | grep -v "/cpu_") /tmp/hand_mw_timing.o -o $ROOT/ab/lib_timing_hmw.so
echo built $ROOT/ab/lib_timing_hmw.so
