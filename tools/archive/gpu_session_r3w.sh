set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_kinematics_views.py tests/test_gymapi_shim.py -q -m gpu -k "jacobian or kinematics or views" > gpurun_out/r3w/pytest_kin.log 2>&1; tail -30 gpurun_out/r3w/pytest_kin.log
