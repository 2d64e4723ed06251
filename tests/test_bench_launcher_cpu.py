"""bench.py's own multi-rank launcher (the role main.py:102-116 plays in the reference), exercised on CPU over gloo:
`python bench.py --gpus N` with no WORLD_SIZE must start N ranks itself, prove the communicator has N ranks, and refuse
to report an N-GPU number from fewer."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_DIST_BACKEND="gloo", BENCH_SINGLE_DEVICE="1", OMP_NUM_THREADS="1", **kw)
    return env


def test_plain_command_self_launches_n_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout                        # rank 0 only
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks"] == 2 and len(lines[0]["devices"]) == 2


def test_plain_command_self_launches_eight_ranks():
    """The driver's 8-GPU line in miniature: eight ranks over gloo, the communicator proven to have eight, devices gathered."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 8 and lines[0]["ranks"] == 8 and len(lines[0]["devices"]) == 8


def test_refuses_a_rank_count_that_differs_from_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")]


def test_single_rank_needs_no_process_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["ranks"] == 1
