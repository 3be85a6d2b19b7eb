#!/bin/bash
# Does a second resident workgroup per CU speed the hand sub-step up?  Two builds of kernels_shadow_hand.hip with KMAX = 3 (row store 626
# slots = 78 KB per workgroup: two fit a CU's 160 KB), one of them padded to 100 KB (one per CU) -- same instructions, same work,
# only the residency differs.  Run here (build) then on the GPU: MI_ENGINE_LIB=ab/lib_hand_k3{,_pad}.so python tools/hand_residency_ab.py
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
B=$ROOT/isaacgymenvs_amd/csrc/build
cd $ROOT/isaacgymenvs_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-trapping-math -fno-slp-vectorize"
mkdir -p $ROOT/ab
hipcc $FLAGS -DMI_HAND_KMAX=3 -c kernels_shadow_hand.hip -o /tmp/hand_k3.o 2>/dev/null &
hipcc $FLAGS -DMI_HAND_KMAX=3 -DMI_HAND_LDS_PAD=24576 -c kernels_shadow_hand.hip -o /tmp/hand_k3_pad.o 2>/dev/null &
wait
OTHERS=$(ls $B/*.o | grep -v "kernels_shadow_hand.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/hand_k3.o -o $ROOT/ab/lib_hand_k3.so
hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/hand_k3_pad.o -o $ROOT/ab/lib_hand_k3_pad.so
echo built
