"""bench.py's own plumbing on the CPU: argument handling, the self-launch of N ranks (`--gpus 2` without a launcher starts
`python -m torch.distributed.run` as a child), the row partition + gloo gather, and the JSON line's contract fields.  The renders
come from the CPU emulator of the kernel source (RTC_BENCH_CPU_STANDIN=1: a rehearsal of the plumbing, never a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None):
    env = dict(os.environ, RTC_BENCH_CPU_STANDIN="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: two ranks are started by bench.py itself, rank 0 prints the one line."""
    j = run_bench("--gpus", "2", "--workload", "smoke", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    # the N-rank line proves its frame too: gathered rows == the whole frame rendered by one rank
    assert j["parity"]["gathered_frame_equals_single_gpu_frame"] is True and j["parity"]["pixels"] == 64 * 36
    assert j["config"]["process_group"] == {"world_size": 2, "backend": "gloo"}
    assert j["config"]["frames_in_flight"] == 3 and j["scaling"] == "strong" and j["value"] > 0
    assert "STAND-IN" in j["data"]
    assert j["config"]["partition"].startswith("bands") and "band_partition" not in j     # 3 frames < 2 N: every frame cut into bands
    # enough frames queued (>= 2 N): whole frames round-robin over the ranks, the band partition measured beside it
    f = run_bench("--gpus", "2", "--workload", "smoke", "--steps", "5", "--warmup", "1", "--no-cpu-baseline")
    assert f["config"]["partition"].startswith("frames") and f["steps"] == 5 and f["n_gpus"] == 2
    assert f["parity"]["gathered_frame_equals_single_gpu_frame"] is True
    assert f["band_partition"]["parity"]["gathered_frame_equals_single_gpu_frame"] is True and f["band_partition"]["value"] > 0
    assert f["config"]["unique_rays_per_frame"] == j["config"]["unique_rays_per_frame"]
    one = run_bench("--gpus", "1", "--workload", "smoke", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--extra-workloads", "")
    # same frame, same unique rays, whatever the partition
    assert one["config"]["unique_rays_per_frame"] == j["config"]["unique_rays_per_frame"]
    assert one["config"]["frames_in_flight"] == j["config"]["frames_in_flight"]


def test_bench_line_contract_fields_single_rank():
    j = run_bench("--workload", "smoke", "--steps", "2", "--warmup", "1", "--extra-workloads", "")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "ms_per_step_incl_d2h"):
        assert k in j, k
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    a = r["algorithmic_bytes"]
    c = r["counters_rank0"]
    # the figure is recomputable from the line's own counters; kernel-argument-resident records are not in it
    want = sum(a["by_unit"].values())
    assert a["counted"] == want and a["memory"] == min(want, a["cap_4x_ideal"]) == r["algorithmic_bytes_per_launch"]
    assert a["by_unit"]["node"] + (a["lds"] and 128 * (c["accel_nodes"] - c["accel_nodes_kernarg"])) == 128 * (c["accel_nodes"] - c["accel_nodes_kernarg"])
    assert a["by_unit"]["analytic"] + (a["lds"] and 128 * (c["analytic_tests"] - c["analytic_tests_kernarg"])) == 128 * (c["analytic_tests"] - c["analytic_tests_kernarg"])
    assert a["lds"] == 0      # the emulator reports no LDS-resident tables
    assert c["accel_nodes_kernarg"] > 0 and c["analytic_tests_kernarg"] > 0      # this scene's planes and BVH root travel in the kernel arguments
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    p = j["parity"]
    assert p["primary_hit_mismatches"] == 0 and p["max_abs_drgb"] <= 1e-5 and p["pixels"] > 0 and p["timed_frame_equals_checked_frame"]
