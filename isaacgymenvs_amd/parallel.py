"""Multi-GPU sharding: one process per GPU, env shards are independent, RCCL moves only episode statistics.

The reference has no collective in its own tree: with `multi_gpu=True` each rank builds its own env
shard on cuda:LOCAL_RANK (utils/rlgames_utils.py:89-107) and rl_games all-reduces gradients.  For the env engine
the only cross-rank quantity is the episode-statistics vector the fused step kernel accumulates
("episode_stats": sum finished returns, sum finished lengths, #finished, sum rewards, #env-steps), summed with one
tiny all-reduce (<= 32 B payload, latency bound) on a side stream every `interval` steps -- never on the step stream.
The tasks' own `extras` travel the same way: AnymalTerrain's `extras["episode"]` (13 episode sums + terrain level + count,
anymal_terrain.py:421-425) and the hand tasks' consecutive-successes numerator / denominator (shadow_hand.py:795-798) are kept by the kernels
as CUMULATIVE sums (`episode_cum_stats`, `reward_workspace[2:4]`), so a window's job-wide means are differences of two all-reduced snapshots
-- no per-step torch op on the step stream (TaskExtrasReducer).
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
            # (MI_DIST_BACKEND: tests only -- two ranks of the GPU path on ONE device need gloo, RCCL refuses a device twice)
            backend = os.getenv("MI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        pin_to_local_cores(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _cores_in_topology_order(cores):
    """the allowed cpus grouped by NUMA node (/sys/devices/system/node/node*/cpulist), nodes in order, so that a contiguous slice is a slice of
    one socket even where SMT siblings are numbered N .. 2N-1 (raw cpu ids would put ranks 4-7 on the hyperthreads of ranks 0-3); falls back to
    the ids as given"""
    import glob
    allowed, out = set(cores), []
    try:
        nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda p: int(p.rsplit("node", 1)[1]))
        for nd in nodes:
            ids = []
            for part in open(os.path.join(nd, "cpulist")).read().strip().split(","):
                if not part:
                    continue
                a, _, b = part.partition("-")
                ids += list(range(int(a), int(b or a) + 1))
            out += [c for c in ids if c in allowed]
    except (OSError, ValueError):
        return cores
    return out if len(out) == len(cores) else cores


def pin_to_local_cores(local_rank=None, local_world=None):
    """One process per GPU: give every rank of a node its own contiguous slice of the cores the job may use (sched_setaffinity).  The engine's host
    side is a single Python thread issuing one launch per step (~25 us of work per 40-170 us step); eight such threads migrating over two sockets
    contend for nothing but each other's caches and the launch path's locks, and a contiguous slice is also the socket next to the GPU on the usual
    board layout (GPUs 0-3 on socket 0, 4-7 on socket 1).  MI_NO_AFFINITY=1 leaves the scheduler alone.  Returns the cores chosen, or None."""
    if os.getenv("MI_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    local_rank = int(os.getenv("LOCAL_RANK", "0")) if local_rank is None else local_rank
    if local_world is None:
        # only the launcher knows how many ranks share this node: without LOCAL_WORLD_SIZE (mpirun / slurm launches) WORLD_SIZE would be the
        # JOB's size and every rank of a multi-node job would get cores / world of its node -- leave the scheduler alone instead
        if "LOCAL_WORLD_SIZE" not in os.environ:
            return None
        local_world = int(os.environ["LOCAL_WORLD_SIZE"])
    if local_world <= 1:
        return None
    cores = _cores_in_topology_order(sorted(os.sched_getaffinity(0)))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def shard_range(num_envs_global, rank, world):
    """Global env ids [lo, hi) owned by `rank` for a strong-scaling split of a fixed global env count."""
    per = num_envs_global // world
    rem = num_envs_global % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


class EpisodeStatsReducer:
    """Periodic SUM all-reduce of the engine's episode-statistics vector on a side stream."""

    FIELDS = ("sum_episode_return", "sum_episode_length", "num_episodes", "sum_reward", "num_env_steps")

    def __init__(self, stats_tensor: torch.Tensor, interval: int = 16, distributed: bool = True):
        self.stats = stats_tensor
        self.interval = max(1, int(interval))
        self.n = 0
        self.dist = bool(distributed) and dist.is_initialized()        # False: this process's statistics only (a reference run inside a job)
        self.world = dist.get_world_size() if self.dist else 1
        self.on_gpu = stats_tensor.is_cuda
        self.side = torch.cuda.Stream(device=stats_tensor.device) if self.on_gpu else None
        self.global_stats = self._snapshot_like()
        self._work = None

    def _snapshot_like(self):
        return torch.zeros_like(self.stats)

    def _snapshot(self):
        """what is all-reduced: a copy of the statistics tensor (subclasses widen it)"""
        self.global_stats.copy_(self.stats)

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
                self._snapshot()
                if self.dist:
                    self._work = dist.all_reduce(self.global_stats, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._snapshot()
            if self.dist:
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


class TaskExtrasReducer(EpisodeStatsReducer):
    """Job-wide task `extras` of a sharded run: SUM all-reduce of the task's cumulative statistics every `interval` steps (side stream), the
    window's means from the difference to the previous reduced snapshot.

      AnymalTerrain  extras["episode"] as the reference forms it per step (anymal_terrain.py:421-425), over the window and over all ranks:
                     rew_<term> = sum of the episode sums of the envs that reset / their number / max_episode_length_s,
                     terrain_level = mean level over envs and steps.  Tensor `episode_cum_stats` [32] (16 compensated sums: high parts, then low parts).
      ShadowHand / AllegroHand   the consecutive-successes average's numerator and denominator (shadow_hand.py:795-798):
                     mean successes of the episodes that ended in the window, and the job-wide moving average
                     cs <- av_factor * that + (1 - av_factor) * cs, updated once per window (per step on a single rank).  `reward_workspace[2:4]` (+ low parts `[4:6]`).
    """

    ANYMAL_KEYS = ("rew_lin_vel_xy", "rew_lin_vel_z", "rew_ang_vel_z", "rew_ang_vel_xy", "rew_orient", "rew_torques", "rew_joint_acc", "rew_base_height",
                   "rew_air_time", "rew_collision", "rew_stumble", "rew_action_rate", "rew_hip")

    def __init__(self, env, interval: int = 16, distributed: bool = True):
        t = env.engine.tensors
        self.task = env.native_task
        if self.task == "AnymalTerrain":
            src = t["episode_cum_stats"]
        elif self.task in ("ShadowHand", "AllegroHand"):
            src = t["reward_workspace"]
        else:
            raise ValueError(f"{self.task}: no task extras to reduce (its episode statistics go through EpisodeStatsReducer)")
        super().__init__(src, interval, distributed)
        self.env = env
        self.prev = torch.zeros_like(self.global_stats, device="cpu")
        self.base = self.prev.clone()
        self.window = None
        self.consecutive_successes = 0.0
        self._launched = 0          # reductions started / turned into a window: a window is formed ONCE per completed reduction
        self._consumed = 0

    # The kernels keep each cumulative sum as a compensated pair of floats (core/rng.hpp two_sum_acc: AnymalTerrain [k] + [16 + k], hands
    # [2], [3] + [4], [5]); the snapshot that is all-reduced is their float64 sum, so neither the accumulation nor the reduction over the ranks
    # loses the small per-window increments once the totals are large (a float32 total passes 2^24 after ~1k steps of 4096 envs).
    def _snapshot_like(self):
        return torch.zeros(16 if self.task == "AnymalTerrain" else 4, dtype=torch.float64, device=self.stats.device)

    def _snapshot(self):
        s = self.stats.double()
        if self.task == "AnymalTerrain":
            torch.add(s[:16], s[16:32], out=self.global_stats)
        else:
            self.global_stats.zero_()
            self.global_stats[2:4] = s[2:4] + s[4:6]

    def reduce_async(self):
        # the previous reduction becomes a window before its buffer is overwritten (it has had `interval` steps to complete: the wait is free)
        if self._launched > self._consumed:
            self._finish_window()
        super().reduce_async()
        self._launched += 1

    def rebase(self):
        """Forget the snapshot the next window is measured from: the cumulative sums were replaced under the reducer (set_env_state restored
        an arena), so a difference to the old snapshot means nothing.  The next reduction only re-establishes the baseline."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._consumed = self._launched
        self.prev = None

    def poll(self):
        """the newest completed window, without blocking: None until the first reduction after the baseline has finished"""
        if self._launched > self._consumed and (self._work is None or self._work.is_completed()) and (
                not self.on_gpu or self.side.query()):
            self._finish_window()
        return self.window

    def result(self):
        """job-wide extras of the last reduced window (blocking).  Calling it again before the next reduction returns the same window."""
        if self._launched > self._consumed:
            self._finish_window()
        return self.window

    def _finish_window(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.on_gpu:
            self.side.synchronize()
        self._consumed = self._launched
        cur = self.global_stats.detach().cpu()
        if self.prev is None or bool((cur[[13, 15]] < self.prev[[13, 15]]).any() if self.task == "AnymalTerrain" else (cur[2] < self.prev[2])):
            self.prev = cur.clone()         # a baseline (after rebase(), or counters that ran backwards): no window from it
            self.base = cur.clone()
            return
        d = cur - self.prev
        self.prev = cur.clone()
        self.window = self._window_of(d, update_average=True)

    def since_baseline(self):
        """job-wide extras over everything reduced since the baseline (construction, or the last rebase()): the same quantities as a window's"""
        if self._launched > self._consumed:
            self._finish_window()
        if self.prev is None:
            return None
        return self._window_of(self.prev - self.base, update_average=False)

    def _window_of(self, d, update_average):
        if self.task == "AnymalTerrain":
            cnt, steps = float(d[13]), max(float(d[15]), 1.0)          # [15] counts every rank's steps: steps of the window x ranks
            out = {}
            if cnt > 0:
                for k, name in enumerate(self.ANYMAL_KEYS):
                    out[name] = float(d[k]) / cnt / float(self.env.max_episode_length_s)
            out["terrain_level"] = float(d[14]) / (self.env.num_envs * steps)        # mean over the envs of every rank and the window's steps
            out["num_resets"] = cnt
            return out
        resets, fin = float(d[2]), float(d[3])
        if resets > 0 and update_average:
            # the engine folds every step's finished episodes into its moving average (shadow_hand.py:795-798); job-wide the same update is made
            # once per window with the window's ratio -- it feeds only `extras`
            av = float(self.env._task_params_struct.rew.av_factor)
            self.consecutive_successes = av * fin / resets + (1.0 - av) * self.consecutive_successes
        return {"num_resets": resets, "successes_per_reset": (fin / resets) if resets > 0 else 0.0,
                "consecutive_successes": self.consecutive_successes}
