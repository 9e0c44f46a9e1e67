"""BASELINE configs[4]'s arithmetic + wire format at two ranks on one GPU (gloo): see tests/dp_cfg5_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("cfgname", ["cfg1", "cfg2"])
def test_cfg5_bf16_class_with_bf16_wire_follows_the_fp32_data_parallel_step(cfgname):
    """cfg2 = BASELINE configs[4]'s shard shape itself (the cfg2 shard had never gone through
    GradAllReduce(comm_dtype=bf16)); cfg1 = the same arithmetic at the reference-plumbing shape."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", RBR_TEST_CFG=cfgname)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541" if cfgname == "cfg1" else "29543", os.path.join(HERE, "dp_cfg5_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "CFG5 BF16 DP OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
