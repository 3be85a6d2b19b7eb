"""bench.py's reporting helpers (no GPU): the roofline / cpu_baseline objects carry the fields the measurement contract names."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_roofline_object_fields_and_arithmetic():
    r = bench.roofline("Ant", 4096, 0.075)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # achieved = algorithmic bytes per env-step (673 B for Ant) x envs / kernel time
    assert r["algorithmic_bytes_per_launch"] == 673 * 4096
    assert abs(r["achieved"] - 673 * 4096 / 0.075e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    # traffic / valu come from profiles/traffic.json (tools/summarize_profile.py): measured PMC values of the same command, or null
    if r["traffic"] is not None:
        assert r["traffic"] > r["algorithmic_bytes_per_launch"]        # PMC traffic includes the per-sub-step re-reads
        assert r["traffic_source"]
    if "valu" in r:
        assert 0 < r["valu"]["frac"] < 1
    assert "multi-wave" not in r["note"] and "4 waves" in bench.roofline("Ant", 4096, 0.05, mw=16)["note"]
    for task, n in bench.DEFAULT_ENVS.items():
        assert bench.roofline(task, n, 1.0)["algorithmic_bytes_per_launch"] == bench.ALGO_BYTES[task] * n


def test_cpu_baseline_reports_the_threads_it_used():
    r = bench.cpu_baseline("Ant", 64, budget_s=1.2)
    assert r["kind"] == "port" and r["unit"] == "env-steps/s" and r["value"] > 0
    assert str(r["cores"]) in r["thread_sweep"] and r["value"] == max(r["thread_sweep"].values()) or abs(r["value"] - max(r["thread_sweep"].values())) < 1.0
    assert r["cores"] <= r["host_threads"]


def test_reference_jit_leg_runs_where_the_reference_is_reachable():
    """the reference's own jitted obs / reward functions on torch-CPU; None / absent elsewhere (e.g. on the GPU box)."""
    leg = bench.reference_jit_leg("Ant", 256, budget_s=0.3)
    if os.path.isdir("/root/reference/isaacgymenvs"):
        assert leg["kind"] == "reference" and leg["value"] > 0 and "compute_ant_observations" in leg["sample"]
    else:
        assert leg is None or "absent" in leg
