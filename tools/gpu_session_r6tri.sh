#!/bin/bash
# round 6: PPO on the reference's UNMODIFIED trifinger.py through the stand-in on the HIP backend (a learnability check of the scene as built:
# fingers, table plate, free cube; the ring wall has no collision shape) and its step time
out=$GRAFT_REPO_ROOT/gpurun_out/r6tri; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python examples/train_ppo.py --task Trifinger --reference-task trifinger:Trifinger --num-envs 4096 --iters 300 --horizon 16 > $out/ppo_trifinger.log 2>&1; echo "ppo rc=$?"; tail -4 $out/ppo_trifinger.log | cut -c1-300
