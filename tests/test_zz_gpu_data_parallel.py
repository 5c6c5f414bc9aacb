"""Two ranks on the one GPU (gloo rendezvous, buckets reduced through a host copy): the whole data-parallel path on CUDA
tensors -- broadcast of rank 0's weights, flat gradient buffer, one all-reduce per finished bucket during backward, fused
Adam.  Collected last: it starts two more GPU processes."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "dp_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("data-parallel gradients and parameters agree") == 2
    assert r.stdout.count("UNetVgg all-reduced gradients match the shard-averaged oracle") == 2


def test_rccl_branch_single_rank():
    """The `nccl` (= RCCL) branch of the bucketed all-reduce, executed for real with a one-rank process group."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_worker.py")], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL branch ran" in r.stdout
