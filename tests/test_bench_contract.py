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
    # the compute-side peaks are the guide's: SIMD-32, a wave64 VALU instruction per 2 cycles -> 256 x 4 x 2.4 GHz / 2 wave-instructions/s,
    # which x 64 lanes x 2 FLOP is the 157.3 TFLOP/s fp32 vector spec (the line used half of it)
    assert abs(bench.VALU_PEAK_GINST - 1228.8) < 1e-9
    assert abs(bench.VALU_PEAK_GINST * 1e9 * 64 * 2 / 1e12 - bench.FP32_PEAK_TFLOPS) < 0.1
    if "fp32" in r:
        assert r["fp32"]["peak"] == 157.3 and 0 < r["fp32"]["frac"] < 1
        assert abs(r["fp32"]["achieved"] - r["fp32"]["flops_per_env_step"] * 4096 / 0.075e-3 / 1e12) < 1e-9
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


def test_every_leg_is_checked_for_kernel_time_inside_step_time():
    """a leg whose HIP-event kernel time exceeds the wall-clock step it is quoted against is an early-episode artefact of a
    too-short run; measure() flags it (`consistent`) and the side legs run at least 200 steps after 50 warm-ups."""
    ok = dict(kernel_ms_avg=0.384, ms_per_step=0.393, pooled=dict(ms_per_step=0.397))
    bad = dict(kernel_ms_avg=0.395, ms_per_step=0.345, pooled=dict(ms_per_step=0.34))
    assert bench.leg_consistent(ok) and not bench.leg_consistent(bad)
    import inspect
    src = inspect.getsource(bench.main)
    assert "max(args.steps // 4, 200)" in src and "max(args.warmup // 4, 50)" in src
    assert 'res["consistent"]' in inspect.getsource(bench.measure)
    # committed bench lines of this round carry the flag on every leg, and it holds
    import glob
    import json
    for f in glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r3*_bench.json")):
        with open(f) as fh:
            line = json.loads(fh.read().strip().splitlines()[-1])
        legs = [line] + [line[k] for k in ("extra", "extra2", "extra3") if k in line]
        for leg in legs:
            assert leg.get("consistent") is True, (f, leg.get("workload", "headline"))
            assert leg["roofline"]["kernel_ms"] * 0.9 <= leg["ms_per_step"]


def test_cpu_product_backend_leg_runs_through_the_public_api():
    r = bench.cpu_product_backend("Ant", 64, budget_s=0.6)
    assert "absent" not in r, r
    assert r["threads_4"]["value"] > 0 and r["kind"] == "product_cpu_backend"
