"""bench.py end to end on the GPU with the driver's own flags (small shard, no CPU baseline): the JSON line carries both
schedules, the shards, the roofline of the timed region's env kernel — and the run comes back."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_with_the_drivers_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--envs-per-gpu", "512", "--global-envs", "1024", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"].startswith("env steps/sec") and line["unit"] == "env steps/s" and line["n_gpus"] == 1
    assert line["steps"] == 20 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["schedule"] in ("pipelined", "synchronous")
    other = "synchronous" if line["schedule"] == "pipelined" else "pipelined"
    assert other in line and "error" not in line[other] and line[other]["value"] > 0
    assert line["value"] >= line[other]["value"]                       # value = the faster schedule's
    assert abs(line["value"] - 20 * 512 / (line["ms_per_step"] * 20 * 1e-3)) / line["value"] < 1e-6
    assert line["timed_gpu_seconds"] >= 0.5 and line["repeats_run"] >= 5
    sizes = line["shards"]["sizes"]
    assert set(sizes) == {"128", "256", "512"} and all(v["value"] > 0 for v in sizes.values())
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["in_timed_region"] is True and 0.0 < roof["frac"] < 1.0
    assert roof["variant"] == "act_step" and "k_act_step" in roof["kernel"]
    # SURVEY 8(d)'s figure: 709 B per env-step (byte observations) over the kernel's in-situ duration; the wider byte count
    # (policy state the fused kernel also moves) only under its own name
    assert roof["bytes_per_env_step"] == 709 and roof["frac"] == roof["frac_8d"]
    assert roof["in_situ"] is not None and roof["avg_launch_us"] > 0 and roof["alone_us"] > 0
    assert roof["policy_state_included"]["bytes_per_env_step"] > 709
    assert "step_f32" in roof["other_variants"] and line["ranks"] == 1
    assert line["config"]["schedule"] == line["schedule"] and "workload" in line["config"]


@pytest.mark.gpu
def test_bench_watchdog_prints_the_synchronous_line_when_the_pipelined_part_does_not_come_back():
    """--pipelined-timeout far below what the pipelined measurements need: the watchdog fires, rank 0 prints the line as
    assembled so far (the synchronous numbers) and the process leaves with exit code 0 — an N>1 run must never hang the driver."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--envs-per-gpu", "512", "--no-shards", "--no-cpu-baseline", "--pipelined-timeout", "0.01"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["schedule"] == "synchronous" and line["value"] > 0
    assert "did not complete" in line["pipelined"]["error"]
    assert line["roofline"]["in_timed_region"] is True
