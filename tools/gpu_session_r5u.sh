#!/bin/bash
# round 5: PPO sanity run of the reference's unmodified franka_cube_stack.py on the scene engine (FrankaCubeStackPPO.yaml: horizon 32, minibatch 16384,
# 5 mini-epochs, lr 5e-4, reward scale 1)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python examples/train_ppo.py --reference-task franka_cube_stack:FrankaCubeStack --task FrankaCubeStack --num-envs 4096 --iters ${1:-150} --horizon 32 \
  --minibatch 16384 --epochs 5 --lr 5e-4 --reward-scale 1.0 > $OUT/ppo_franka.log 2>&1; echo "ppo rc=$?"; grep '"iter"' $OUT/ppo_franka.log | tail -16 | cut -c1-250
