"""-m gpu, needs >= 2 visible GPUs (skipped on the 1-GPU box): the N>1 product path on RCCL itself — what the world-size-2
gloo test (tests/test_distributed_cpu.py) checks on CPU, here with one rank per GPU over xGMI: all-reduced gradient == mean
of the ranks' gradients, three hipGraph-replayed iterations with the eager all-reduce between the graphs, replicas
bit-identical afterwards. Replaces utils.py:36-44 + shared_optim.py:113-120 of the reference."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nranks):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_worker.py")],
                       env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0 and ("RCCL_OK world=%d" % nranks) in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.timeout(600)
def test_two_rank_rccl_graphed_iterations_keep_replicas_identical():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    _run(2)


@pytest.mark.timeout(600)
def test_one_rank_rccl_process_group_runs_the_same_worker():
    """The same worker under torch.distributed.run with ONE rank: brings up the RCCL process group and runs the launcher,
    shard construction and graphed iterations on the 1-GPU box (the cross-rank assertions need the test above)."""
    _run(1)
