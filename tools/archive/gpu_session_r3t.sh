set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3t/pytest_gpu.log 2>&1; tail -5 gpurun_out/r3t/pytest_gpu.log
timeout 600 bash tools/ab_bench.sh ab/lib_head_3d48db7.so isaacgymenvs_amd/libmi_engine.so > gpurun_out/r3t/ab_head_vs_scaled_split.txt 2>&1; cat gpurun_out/r3t/ab_head_vs_scaled_split.txt
(echo "== this build"; timeout 300 python tools/actor_tensors_ab.py; echo "== lib_head_3d48db7 (factors inside the one kernel)"; MI_ENGINE_LIB=$PWD/ab/lib_head_3d48db7.so timeout 300 python tools/actor_tensors_ab.py) > gpurun_out/r3t/actor_tensors_ab.txt 2>&1; cat gpurun_out/r3t/actor_tensors_ab.txt
