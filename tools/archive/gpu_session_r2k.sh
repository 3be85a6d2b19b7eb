#!/bin/bash
# PPO plausibility runs of the tasks whose physics is new in round 2 (examples/train_ppo.py, hyper-parameters of the reference's cfg/train/*PPO.yaml)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 400 python examples/train_ppo.py --task BallBalance --iters 250 --units 128,64,32 --reward-scale 0.1 --minibatch 8192 --epochs 8 --log $OUT/ppo_ballbalance.json > $OUT/ppo_ballbalance.txt 2>&1; tail -2 $OUT/ppo_ballbalance.txt
timeout 600 python examples/train_ppo.py --task Ingenuity --iters 500 --units 256,256,128 --lr 1e-3 --minibatch 16384 --epochs 8 --log $OUT/ppo_ingenuity.json > $OUT/ppo_ingenuity.txt 2>&1; tail -2 $OUT/ppo_ingenuity.txt
timeout 900 python examples/train_ppo.py --task Humanoid --iters 1000 --units 400,200,100 --log $OUT/ppo_humanoid.json > $OUT/ppo_humanoid.txt 2>&1; tail -2 $OUT/ppo_humanoid.txt
