#!/bin/bash
# ab/lib_timing.so = the current objects with kernels_humanoid.hip and kernels_shadow_hand.hip rebuilt with -DMI_TIMING (s_memtime
# stamps per sub-step phase); the Humanoid then runs its one-wave kernel (MI_MULTI_WAVE=0 in tools/debug/phase_timing_live.py)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
B=$ROOT/isaacgymenvs_amd/csrc/build
cd $ROOT/isaacgymenvs_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize"
hipcc $FLAGS -DMI_TIMING -c kernels_humanoid.hip -o /tmp/hum_timing.o 2>/dev/null &
hipcc $FLAGS -DMI_TIMING -c kernels_shadow_hand.hip -o /tmp/hand_timing.o 2>/dev/null &
wait
mkdir -p $ROOT/ab
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v "kernels_humanoid.o\|kernels_shadow_hand.o") /tmp/hum_timing.o /tmp/hand_timing.o -o $ROOT/ab/lib_timing.so
echo built $ROOT/ab/lib_timing.so
