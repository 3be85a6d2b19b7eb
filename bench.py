#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused VecTask.step() hot path on N MI355X (one process per GPU).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
`python -m torch.distributed.run --nproc-per-node N ...` (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from env).
Rank 0 prints ONE JSON line.

  step      = one `env.step(actions)` through the public Python API (isaacgymenvs_amd.make -> VecTask.step), i.e.
              one launch group (sub-step kernels + post kernel, no host sync) advancing every env by one control
              step (reference vec_task.py:360-408).
  workload  = BASELINE.json configs[1]: Ant num_envs=4096 per GPU, random-action rollout (reference README.md:48-51);
              actions come from a pool of pre-generated U(-1,1) batches already resident in HBM.
  value     = (envs on all ranks) * K / max-over-ranks(wall time of the K timed steps), barrier+synchronize on both sides.
  roofline  = algorithmic bytes per control step (673 B/env-step Ant, 1161 B Humanoid) x envs / average
              GPU time of one step's launch group (substep_kernel x substeps + post kernel), measured with HIP events
              around each group on the launch stream (a second pass right after the timed one), against the 8 TB/s
              HBM3E peak.  `traffic` = FETCH_SIZE + WRITE_SIZE of the group from the rocprofv3 PMC passes committed
              under profiles/ (constants below are refreshed from there; null when not measured for the task).
  cpu_baseline = the CPU oracle (oracle/physics.c via oracle/tasks.py, OpenMP over envs) on a bounded sample of the
              same workload on this host's cores ("port": the reference's PhysX-CPU path cannot run).
  extra     = the second headline config (Humanoid num_envs=8192) measured the same way in the same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# API-visible state read once + written once per env-step
ALGO_BYTES = {"Cartpole": 89, "Ant": 673, "Humanoid": 1161, "AnymalTerrain": 2240, "ShadowHand": 3600}
DEFAULT_ENVS = {"Cartpole": 64, "Ant": 4096, "Humanoid": 8192, "AnymalTerrain": 4096, "ShadowHand": 16384}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
# HBM-side bytes per control step from the round-1 PMC passes (profiles/r1_pmc_summary.md): raw FETCH_SIZE + WRITE_SIZE
# (KB -> B) summed over the launches of one step (sub-steps + pre/post kernels) at the BASELINE env counts.  The raw fetch counter
# matches the byte count of our dword-per-lane coalesced loads, so the guide's x2 (calibrated on 16 B/lane streams) is
# NOT applied; the excess over the algorithmic bytes is state re-read per sub-step launch, warm-start impulses and
# (Humanoid) the constraint rows that spill to scratch.
PMC_TRAFFIC_BYTES = {("Ant", 4096): int((2 * (1196.8 + 1856.0) + 661.1 + 2213.1) * 1024),
                     ("Humanoid", 8192): int((2 * (12271.3 + 22448.9) + 2265.5 + 8945.1) * 1024),
                     ("AnymalTerrain", 4096): int((5 * (1623.6 + 2272.0) + 65.0 + 0.1 + 1342.6 + 2549.4 + 740.7 + 4744.1) * 1024),
                     ("ShadowHand", 16384): int((1699.0 + 4689.0 + 2 * (9257.8 + 27090.6) + 7982.1 + 48825.9 + 1.3) * 1024)}


def measure(task, num_envs, steps, warmup, device, rank, world, seed=42, pool=8):
    import torch
    import torch.distributed as dist
    import isaacgymenvs_amd

    env = isaacgymenvs_amd.make(seed=seed + rank, task=task, num_envs=num_envs, sim_device=device, rl_device=device,
                                headless=True, multi_gpu=world > 1, force_render=False)
    g = torch.Generator(device=device).manual_seed(seed + rank)  # reference utils/utils.py:94: seed + rank
    acts = [2.0 * torch.rand((num_envs, env.num_actions), device=device, generator=g) - 1.0 for _ in range(pool)]
    reducer = None
    use_dist = dist.is_available() and dist.is_initialized()
    if use_dist:
        from isaacgymenvs_amd.parallel import EpisodeStatsReducer
        reducer = EpisodeStatsReducer(env.engine.tensors["episode_stats"], interval=16)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        env.step(acts[i % pool])
        if reducer:
            reducer.step()
    sync()
    stats0 = env.engine.tensors["episode_stats"].clone()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        env.step(acts[i % pool])
        if reducer:
            reducer.step()
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if use_dist:
        t = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    stats = (env.engine.tensors["episode_stats"] - stats0).cpu().tolist()
    # per-launch kernel duration: HIP events on the launch stream around each fused-step launch (same workload continuing)
    kn = min(steps, 200)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(kn)]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(pairs):
        a.record()
        env.engine.step(acts[i % pool])
        b.record()
    torch.cuda.synchronize()
    durs = sorted(a.elapsed_time(b) for a, b in pairs)
    kern_ms = sum(durs) / len(durs)
    res = {
        "task": task, "num_envs_per_gpu": num_envs, "wall_s": wall, "ms_per_step": 1e3 * wall / steps,
        "gpu_ms_per_step": gpu_ms / steps, "kernel_ms_avg": kern_ms, "kernel_ms_median": durs[len(durs) // 2],
        "env_steps_per_s": world * num_envs * steps / wall,
        "reset_rate": stats[2] / max(stats[4], 1.0), "mean_reward": stats[3] / max(stats[4], 1.0),
    }
    if reducer:
        res["job_stats"] = reducer.result()
    del env
    return res


# Instructions one wave executes per control step (rocprofv3 SQ_INSTS_VALU + SALU + LDS per wave, profiles/r1_pmc_summary.md: sub-step
# launches + post kernel) and what a wave that owns its SIMD can issue (tools/debug/ifetch_bench.hip: 4 cycles per 4-byte and ~5.3
# per 8-byte instruction at ~1.8 GHz, ~45 % 8-byte => ~2.55 ns).  Every wave runs concurrently at the BASELINE sizes, so this
# per-wave issue time is the floor of the step on the current one-env-per-lane design; reported next to the mandatory HBM roofline.
WAVE_INSTRS_PER_STEP = {"Ant": 2 * 10800 + 1200, "Humanoid": 2 * 35940 + 2 * 3350}   # Humanoid post: 2 half-filled waves per 64 envs   # SQ_INSTS_VALU + SALU + LDS per wave (profiles/r1_pmc_summary.md)
ISSUE_NS_PER_INSTR = 2.55
WAVE_VALU_PER_STEP = {"Ant": 2 * 9213 + 1104, "Humanoid": 2 * 30808 + 3188, "AnymalTerrain": 5 * 14594 + 2718, "ShadowHand": 2 * 62333 + 1313 + 7549}
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12


def roofline(task, num_envs, kernel_ms):
    bytes_per_launch = ALGO_BYTES[task] * num_envs
    lanes = 32 if task in ("Humanoid", "ShadowHand") else 64   # compact-store models run 32 envs per wave
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = PMC_TRAFFIC_BYTES.get((task, num_envs))
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "kernel": "mi::substep_kernel<%s> (x sim steps) + post kernel = one step" % task,
           "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
           "note": "latency/issue-bound path: %d waves of %d envs, one per SIMD; see DESIGN.md" % ((num_envs + lanes - 1) // lanes, lanes)}
    if task in WAVE_VALU_PER_STEP:
        # SURVEY 8(d) asks for the fp32 side next to the HBM fraction: executed VALU lane-operations (SQ_INSTS_VALU per wave x waves x
        # lanes) against what the chip can issue, 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s (an FMA counts once)
        waves = (num_envs + lanes - 1) // lanes
        lane_ops = WAVE_VALU_PER_STEP[task] * waves * lanes / (kernel_ms * 1e-3)
        out["valu"] = {"achieved": lane_ops / 1e12, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s", "frac": lane_ops / 1e12 / VALU_PEAK_TLANEOPS,
                       "note": "%d of 1024 SIMDs hold a wave at this env count" % waves}
    if task in WAVE_INSTRS_PER_STEP:
        floor_ms = WAVE_INSTRS_PER_STEP[task] * ISSUE_NS_PER_INSTR * 1e-6
        out["single_wave_issue_floor"] = {"instructions_per_wave_per_step": WAVE_INSTRS_PER_STEP[task], "ns_per_instruction": ISSUE_NS_PER_INSTR,
                                          "floor_ms": floor_ms, "frac_of_floor": floor_ms / kernel_ms}
    return out


def cpu_baseline(task, num_envs, budget_s=15.0, seed=42):
    """The CPU oracle on this host's cores (OpenMP over envs), same workload, bounded to ~budget_s seconds."""
    import numpy as np
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.registry import load_model, load_selfcol, sensor_bodies
    from isaacgymenvs_amd.tasks.locomotion import loco_params_from_cfg
    from isaacgymenvs_amd.utils.config import compose
    from oracle.tasks import OracleLocomotionEnv

    cfg = compose(overrides=[f"task={task}"])["task"]
    name = task.lower()
    p = loco_params_from_cfg(cfg, name, 0.44 if task == "Ant" else 1.34)
    px = cfg["sim"]["physx"]
    sim = dict(dt=cfg["sim"]["dt"], substeps=cfg["sim"]["substeps"],
               iters=px["num_position_iterations"] + px["num_velocity_iterations"], gravity=tuple(cfg["sim"]["gravity"]),
               contact_offset=px["contact_offset"], rest_offset=px["rest_offset"],
               max_depen_vel=px["max_depenetration_velocity"], erp=0.5,
               plane_mu=cfg["env"]["plane"]["staticFriction"], ground_z=0.0, cfm=1e-6, warm=1.0)
    sc = load_selfcol(name)        # the Humanoid collides with itself, in the port as in the kernels
    orc = OracleLocomotionEnv(task == "Humanoid", load_model(name), sensor_bodies(name), sim, p, num_envs, seed=seed,
                              precision="f32", **(dict(selfcol=sc, kmax=12, kpair=3) if sc else {}))
    rng = np.random.default_rng(seed)
    nact = orc.nd
    for _ in range(2):
        orc.step(rng.uniform(-1, 1, (num_envs, nact)).astype(np.float32))
    # OpenMP over envs: more threads is not monotonically faster (at 4096 envs a chunk is 16 envs on 256 threads and the numpy obs /
    # reward share is serial), so the port is timed at several thread counts, ~budget_s in total, and the best one is the baseline;
    # 4 threads is also what the reference's CPU pipeline gives PhysX (`num_threads: 4`, cfg/config.yaml:30).
    import ctypes
    all_cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    counts = sorted({c for c in (4, 16, 64, all_cores) if c <= all_cores}) if gomp is not None else [all_cores]
    sweep = {}
    for c in counts:
        if gomp is not None:
            gomp.omp_set_num_threads(c)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / len(counts) and n < 2000:
            orc.step(rng.uniform(-1, 1, (num_envs, nact)).astype(np.float32))
            n += 1
        dt = time.perf_counter() - t0
        sweep[c] = (num_envs * n / dt, n, dt)
    best = max(sweep, key=lambda c: sweep[c][0])
    out = {"value": sweep[best][0], "unit": "env-steps/s", "cores": best, "kind": "port",
           "sample": f"{sweep[best][1]} steps of {task} num_envs={num_envs} (oracle/physics.c fp32, OpenMP over envs + numpy obs/reward), "
                     f"{sweep[best][2]:.1f} s on {best} threads (best of the sweep); stand-in for PhysX-CPU, which cannot run here",
           "thread_sweep": {str(c): round(v[0]) for c, v in sweep.items()}, "host_threads": all_cores}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--task", default="Ant", choices=list(DEFAULT_ENVS))
    ap.add_argument("--num-envs", type=int, default=0, help="envs per GPU (default: the BASELINE config of the task)")
    ap.add_argument("--no-extra", action="store_true", help="skip the Humanoid@8192 side measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--pool", type=int, default=8,
                    help="number of pre-generated U(-1,1) action batches cycled through.  The reference protocol draws the actions with "
                         "torch.rand right before every step (README.md:48-51), i.e. they are cache-hot when step() reads them; a small "
                         "pool reproduces that without timing torch.rand (a pool of 64 distinct batches made every read a cold miss, +5 %)")
    args = ap.parse_args()

    import torch
    from isaacgymenvs_amd.parallel import init_distributed
    rank, world, local_rank = init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU product path)")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    n_env = args.num_envs or DEFAULT_ENVS[args.task]

    main_res = measure(args.task, n_env, args.steps, args.warmup, device, rank, world, pool=args.pool)
    extra = extra2 = extra3 = None
    if not args.no_extra and args.task == "Ant" and world == 1:   # side measurements only in the single-GPU run
        extra = measure("Humanoid", DEFAULT_ENVS["Humanoid"], max(args.steps // 4, 10), max(args.warmup // 4, 5), device, rank, world)
        extra2 = measure("AnymalTerrain", DEFAULT_ENVS["AnymalTerrain"], max(args.steps // 4, 10), max(args.warmup // 4, 5), device, rank, world)
        extra3 = measure("ShadowHand", DEFAULT_ENVS["ShadowHand"], max(args.steps // 8, 10), max(args.warmup // 8, 5), device, rank, world)
    import torch.distributed as dist
    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    out = {
        "metric": "env-steps/sec (num_envs x control-steps/sec), random-action rollout",
        "value": main_res["env_steps_per_s"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.task} num_envs={n_env} per GPU ({world * n_env} total), VecTask.step() via Python API, "
                               f"pool of {args.pool} pre-generated U(-1,1) action batches, seed 42+rank",
                   "task": args.task, "num_envs_per_gpu": n_env, "parallelism": f"env-shard x{world}"},
        "gpu_ms_per_step": main_res["gpu_ms_per_step"], "reset_rate": main_res["reset_rate"],
        "mean_reward": main_res["mean_reward"],
        "roofline": roofline(args.task, n_env, main_res["kernel_ms_avg"]),
    }
    if "job_stats" in main_res:
        out["job_stats"] = main_res["job_stats"]
    if extra is not None:
        out["extra"] = {"workload": f"Humanoid num_envs={DEFAULT_ENVS['Humanoid']} per GPU", "value": extra["env_steps_per_s"],
                        "unit": "env-steps/s", "ms_per_step": extra["ms_per_step"], "reset_rate": extra["reset_rate"],
                        "roofline": roofline("Humanoid", DEFAULT_ENVS["Humanoid"], extra["kernel_ms_avg"])}
    if extra2 is not None:
        out["extra2"] = {"workload": f"AnymalTerrain num_envs={DEFAULT_ENVS['AnymalTerrain']} per GPU (5 sim steps of 5 ms per control step)",
                         "value": extra2["env_steps_per_s"], "unit": "env-steps/s", "ms_per_step": extra2["ms_per_step"],
                         "reset_rate": extra2["reset_rate"],
                         "roofline": roofline("AnymalTerrain", DEFAULT_ENVS["AnymalTerrain"], extra2["kernel_ms_avg"])}
    if extra3 is not None:
        out["extra3"] = {"workload": f"ShadowHand (block, full_state) num_envs={DEFAULT_ENVS['ShadowHand']} per GPU (2 sub-steps per control step)",
                         "value": extra3["env_steps_per_s"], "unit": "env-steps/s", "ms_per_step": extra3["ms_per_step"],
                         "reset_rate": extra3["reset_rate"],
                         "roofline": roofline("ShadowHand", DEFAULT_ENVS["ShadowHand"], extra3["kernel_ms_avg"])}
    if world == 1 and not args.no_cpu_baseline and args.task in ("Ant", "Humanoid"):
        out["cpu_baseline"] = cpu_baseline(args.task, n_env, budget_s=args.cpu_budget)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
