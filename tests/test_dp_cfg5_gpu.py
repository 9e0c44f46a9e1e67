"""BASELINE configs[4]'s arithmetic + wire format at two ranks on one GPU (gloo): see tests/dp_cfg5_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cfg5_bf16_class_with_bf16_wire_follows_the_fp32_data_parallel_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(HERE, "dp_cfg5_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CFG5 BF16 DP OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
