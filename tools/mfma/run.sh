#!/bin/bash
# builds and runs tools/mfma/delassus_mfma_bench.hip on the GPU box; output -> gpurun_out/mfma_delassus.txt
set -e
cd $(dirname $0)/../..
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma/delassus_mfma_bench.hip -o /tmp/delassus_mfma_bench
mkdir -p gpurun_out
/tmp/delassus_mfma_bench 8192 | tee gpurun_out/mfma_delassus.txt
/tmp/delassus_mfma_bench 65536 | tee -a gpurun_out/mfma_delassus.txt
