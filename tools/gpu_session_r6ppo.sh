#!/bin/bash
# round 6, final build: PPO sanity runs on the physics of this round (one friction rule in every engine form, the Humanoid's re-cut limb waves, the
# ShadowHand with the pairs' forces in its fingertip sensors) -- not a parity claim, a check that the tasks are still learnable as built; plus the
# reference's unmodified franka_cube_stack.py (scene engine with the new box contacts)
out=$GRAFT_REPO_ROOT/gpurun_out/r6ppo; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 200 python examples/train_ppo.py --task Ant --iters 400 > $out/ant.log 2>&1; echo "ant rc=$?"; tail -3 $out/ant.log
timeout 300 python examples/train_ppo.py --task Humanoid --num-envs 8192 --iters 600 > $out/humanoid.log 2>&1; echo "humanoid rc=$?"; tail -3 $out/humanoid.log
timeout 300 python examples/train_ppo.py --task AnymalTerrain --iters 400 > $out/anymal_terrain.log 2>&1; echo "anymal rc=$?"; tail -3 $out/anymal_terrain.log
timeout 400 python examples/train_ppo.py --task ShadowHand --num-envs 16384 --iters 1200 --horizon 8 --epochs 5 > $out/shadow_hand.log 2>&1; echo "hand rc=$?"; tail -3 $out/shadow_hand.log
