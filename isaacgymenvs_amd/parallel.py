"""Multi-GPU sharding: one process per GPU, env shards are independent, RCCL moves only episode statistics.

The reference has no collective in its own tree: with `multi_gpu=True` each rank builds its own env
shard on cuda:LOCAL_RANK (utils/rlgames_utils.py:89-107) and rl_games all-reduces gradients.  For the env engine
the only cross-rank quantity is the episode-statistics vector the fused step kernel accumulates
("episode_stats": sum finished returns, sum finished lengths, #finished, sum rewards, #env-steps), summed with one
tiny all-reduce (<= 32 B payload, latency bound) on a side stream every `interval` steps -- never on the step stream.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the torchrun rendezvous if there is one (RANK/WORLD_SIZE/MASTER_* env).  Returns (rank, world, local_rank)."""
    world = int(os.getenv("WORLD_SIZE", "1"))
    rank = int(os.getenv("RANK", "0"))
    local_rank = int(os.getenv("LOCAL_RANK", "0"))
    # MI_FORCE_DIST=1 joins the rendezvous even at world size 1 (exercises the RCCL path on a single-GPU box)
    force = os.getenv("MI_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(num_envs_global, rank, world):
    """Global env ids [lo, hi) owned by `rank` for a strong-scaling split of a fixed global env count."""
    per = num_envs_global // world
    rem = num_envs_global % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


class EpisodeStatsReducer:
    """Periodic SUM all-reduce of the engine's episode-statistics vector on a side stream."""

    FIELDS = ("sum_episode_return", "sum_episode_length", "num_episodes", "sum_reward", "num_env_steps")

    def __init__(self, stats_tensor: torch.Tensor, interval: int = 16):
        self.stats = stats_tensor
        self.interval = max(1, int(interval))
        self.n = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on_gpu = stats_tensor.is_cuda
        self.side = torch.cuda.Stream(device=stats_tensor.device) if self.on_gpu else None
        self.global_stats = torch.zeros_like(stats_tensor)
        self._work = None

    def step(self):
        """Call once per env step; launches the reduction every `interval` steps without blocking the step stream."""
        self.n += 1
        if self.n % self.interval:
            return
        self.reduce_async()

    def reduce_async(self):
        if self._work is not None:  # never two reductions in flight on the same buffer
            self._work.wait()
            self._work = None
        if self.on_gpu:
            self.side.wait_stream(torch.cuda.current_stream(self.stats.device))
            with torch.cuda.stream(self.side):
                self.global_stats.copy_(self.stats)
                if dist.is_initialized():
                    self._work = dist.all_reduce(self.global_stats, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self.global_stats.copy_(self.stats)
            if dist.is_initialized():
                self._work = dist.all_reduce(self.global_stats, op=dist.ReduceOp.SUM, async_op=True)

    def result(self):
        """Blocking read of the last reduced statistics as a dict (job-wide sums and derived means)."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.on_gpu:
            self.side.synchronize()
        v = self.global_stats.detach().cpu().tolist()
        d = dict(zip(self.FIELDS, v[:5]))
        ne = max(d["num_episodes"], 1.0)
        d["mean_episode_return"] = d["sum_episode_return"] / ne
        d["mean_episode_length"] = d["sum_episode_length"] / ne
        d["mean_reward"] = d["sum_reward"] / max(d["num_env_steps"], 1.0)
        return d
