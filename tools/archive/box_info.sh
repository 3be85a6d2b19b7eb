#!/bin/bash
# What distinguishes the boxes of the pool (profiles/r3z_box_probe.txt): system facts next to the device probe and a short timing.
echo "kernel $(uname -r)"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|gfx950|Wavefront|Cacheline|L2:|L3:|Name: +amdgcn" | sort | uniq -c | head -20
for f in noretry vm_fragment_size sched_policy; do [ -r /sys/module/amdgpu/parameters/$f ] && echo "amdgpu.$f=$(cat /sys/module/amdgpu/parameters/$f)"; done
env | grep -E "^HSA_|^HIP_|^ROCR|^GPU_|^AMD_" | sort
rocm-smi --showmemuse --showuse --showxgmierr 2>/dev/null | grep -E "GPU\[" | head -6
rocm-smi -a 2>/dev/null | grep -iE "voltage|temperature.*junction|throttle|pcie|firmware.*(SMC|MEC|SDMA)|vbios" | head -14
nproc; grep -m1 "model name" /proc/cpuinfo
python tools/box_probe.py 2>&1 | grep -v Forcing | head -3 | cut -c1-170
