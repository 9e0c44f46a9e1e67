"""Data-parallel tap exchange of the table gradient: two ranks on one GPU (gloo), see tests/dp_tap_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_tap_exchange_matches_dense_allreduce():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(HERE, "dp_tap_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "TAP EXCHANGE OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
