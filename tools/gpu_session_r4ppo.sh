#!/bin/bash
# round 4, final build: PPO sanity runs (Ant with the whole step in one launch, Humanoid with the post step on its role waves, ShadowHand with the
# reshaped pre / post kernels) -- not a parity claim, a check that the tasks are still learnable as built
out=gpurun_out/r4ppo; mkdir -p $out
timeout 200 python examples/train_ppo.py --task Ant --iters 400 > $out/ant.log 2>&1; echo "ant rc=$?"; tail -3 $out/ant.log
timeout 300 python examples/train_ppo.py --task Humanoid --num-envs 8192 --iters 600 > $out/humanoid.log 2>&1; echo "humanoid rc=$?"; tail -3 $out/humanoid.log
timeout 300 python examples/train_ppo.py --task ShadowHand --num-envs 16384 --iters 1200 --horizon 8 --epochs 5 > $out/shadow_hand.log 2>&1; echo "hand rc=$?"; tail -3 $out/shadow_hand.log
