"""N>1 path on CPU: world_size-2 gloo run of the sharding helpers and the episode-statistics all-reduce."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    from isaacgymenvs_amd.parallel import init_distributed, shard_range, EpisodeStatsReducer
    import torch.distributed as dist
    rank, world, _ = init_distributed(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    lo, hi = shard_range(4097, rank, world)
    sizes = [None, None]
    dist.all_gather_object(sizes, (lo, hi))
    assert sizes[0][0] == 0 and sizes[0][1] == sizes[1][0] and sizes[1][1] == 4097
    stats = torch.zeros(8)
    red = EpisodeStatsReducer(stats, interval=4)
    for step in range(8):
        stats[0] += 10.0 * (rank + 1); stats[1] += 100.0; stats[2] += 1.0; stats[3] += 0.5; stats[4] += 64
        red.step()
    r = red.result()
    assert r["num_episodes"] == 16 and r["sum_episode_return"] == 8 * 10 + 8 * 20 and r["num_env_steps"] == 2 * 8 * 64, r
    assert abs(r["mean_episode_return"] - 15.0) < 1e-6 and r["mean_episode_length"] == 100.0
    dist.barrier()
    if rank == 0:
        print("GLOO_OK")
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK" in outs[0]
