#!/bin/bash
# Stages the FEW reference files tests/test_gymapi_shim.py imports under ab/ref_stage (git-ignored: they travel to the GPU box with a
# gpurun snapshot and are never committed) so that the shim tests can run the reference's unmodified task files on the HIP backend.
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
REF=${1:-/root/reference}
D=$ROOT/ab/ref_stage/isaacgymenvs
rm -rf $ROOT/ab/ref_stage
mkdir -p $D/tasks/base $D/utils $D/cfg
for f in cartpole ant humanoid anymal_terrain shadow_hand allegro_hand anymal ball_balance quadcopter ingenuity franka_cube_stack; do cp $REF/isaacgymenvs/tasks/$f.py $D/tasks/; done
cp $REF/isaacgymenvs/tasks/base/vec_task.py $D/tasks/base/
cp $REF/isaacgymenvs/utils/*.py $D/utils/
cp -r $REF/isaacgymenvs/cfg/. $D/cfg/
mkdir -p $ROOT/ab/ref_stage/assets/mjcf && cp $REF/assets/mjcf/nv_ant.xml $ROOT/ab/ref_stage/assets/mjcf/     # tests/test_runtime_assets.py perturbs a copy
mkdir -p $ROOT/ab/ref_stage/assets/urdf/anymal_c/urdf && cp $REF/assets/urdf/anymal_c/urdf/anymal.urdf $ROOT/ab/ref_stage/assets/urdf/anymal_c/urdf/   # anymal.py:168: parsed, its tree checked
# tests/test_articulation.py: humanoid_amp.py with its base class, motion library and poselib (code only), the robot and one motion clip
cp $REF/isaacgymenvs/tasks/humanoid_amp.py $D/tasks/
mkdir -p $D/tasks/amp/poselib && cp $REF/isaacgymenvs/tasks/amp/*.py $D/tasks/amp/ && cp -r $REF/isaacgymenvs/tasks/amp/utils_amp $D/tasks/amp/
cp -r $REF/isaacgymenvs/tasks/amp/poselib/poselib $D/tasks/amp/poselib/ && cp $REF/isaacgymenvs/tasks/amp/poselib/*.py $D/tasks/amp/poselib/ 2>/dev/null || true
cp $REF/assets/mjcf/amp_humanoid.xml $ROOT/ab/ref_stage/assets/mjcf/
mkdir -p $ROOT/ab/ref_stage/assets/amp/motions && cp $REF/assets/amp/motions/amp_humanoid_run.npy $ROOT/ab/ref_stage/assets/amp/motions/
# tests/test_gymapi_shim.py: the dextreme task (allegro_hand_dextreme.py + adr_vec_task.py)
mkdir -p $D/tasks/dextreme && cp $REF/isaacgymenvs/tasks/dextreme/*.py $D/tasks/dextreme/
# tests/test_articulation.py: the Franka arm of franka_cube_stack.py:189 (URDF + its collision meshes)
mkdir -p $ROOT/ab/ref_stage/assets/urdf/franka_description/robots $ROOT/ab/ref_stage/assets/urdf/franka_description/meshes/collision
cp $REF/assets/urdf/franka_description/robots/franka_panda_gripper.urdf $ROOT/ab/ref_stage/assets/urdf/franka_description/robots/
cp $REF/assets/urdf/franka_description/meshes/collision/*.obj $ROOT/ab/ref_stage/assets/urdf/franka_description/meshes/collision/
# tests/test_scene.py: trifinger.py with its robot, stage and object files (round 6: one-body URDF files as scene actors)
cp $REF/isaacgymenvs/tasks/trifinger.py $D/tasks/
mkdir -p $ROOT/ab/ref_stage/assets/trifinger && cp -r $REF/assets/trifinger/robot_properties_fingers $REF/assets/trifinger/objects $ROOT/ab/ref_stage/assets/trifinger/
rm -rf $ROOT/ab/ref_stage/assets/trifinger/robot_properties_fingers/meshes/pro/detailed $ROOT/ab/ref_stage/assets/trifinger/robot_properties_fingers/meshes/edu
find $ROOT/ab/ref_stage -name __pycache__ -type d -prune -exec rm -rf {} +
echo staged $(find $ROOT/ab/ref_stage -type f | wc -l) files under ab/ref_stage
# tests/test_articulation.py: the arm + hand of the allegro_kuka tasks (allegro_kuka_base.py:573): the URDF and the COLLISION meshes it names
K=$ROOT/ab/ref_stage/assets/urdf/kuka_allegro_description
mkdir -p $K/meshes/allegro $K/meshes/iiwa7/collision $K/meshes/mounts $K/meshes/touchsensor/collision
cp $REF/assets/urdf/kuka_allegro_description/kuka_allegro_touch_sensor.urdf $K/
cp $REF/assets/urdf/kuka_allegro_description/meshes/allegro/*.obj $K/meshes/allegro/
cp $REF/assets/urdf/kuka_allegro_description/meshes/iiwa7/collision/*.obj $K/meshes/iiwa7/collision/
cp $REF/assets/urdf/kuka_allegro_description/meshes/mounts/*.obj $K/meshes/mounts/
cp $REF/assets/urdf/kuka_allegro_description/meshes/touchsensor/collision/*.obj $K/meshes/touchsensor/collision/
echo staged $(find $ROOT/ab/ref_stage -type f | wc -l) files under ab/ref_stage "(with the kuka_allegro arm + hand)"
