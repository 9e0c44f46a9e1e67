"""Worker of tests/test_dp_cfg5_gpu.py: launched by torch.distributed.run, 2 ranks on ONE GPU over gloo.

BASELINE configs[4] ("DeepCoNN batch 2048 data-parallel across 8xMI355X, RCCL grad all-reduce over xGMI, bf16") as bench.py's
multi-rank `cfg5_bf16` variant runs it -- the conv contraction in the plain-bf16 class (bf16 storage included) and every
gradient all-reduced densely in a bf16 wire format -- against the fp32 data-parallel step (f32-class conv, exact fp32
all-reduce) on the same shards, within the bf16 tolerance class of tests/test_precision_gpu.py.  The reference's semantics:
nn.DataParallel reduces the shard gradients before the clip (trainer/train_deepconn_pp.py:129-131,165-167)."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch
import torch.distributed as dist

import synth
from review_based_recommender_amd import _lib
from review_based_recommender_amd import functional as RF
from review_based_recommender_amd.distributed import GradAllReduce, init_process_group_from_env
from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step


def main():
    init_process_group_from_env("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _lib.lib().rbr_set_conv_mode(2)
    # RBR_TEST_CFG=cfg1: B=32 per rank, L=300, D=100 -- the cfg5 arithmetic at a size two ranks on one card finish in seconds;
    # RBR_TEST_CFG=cfg2: configs[4]'s OWN shard shape (256 pairs per rank, 2x512 tokens, D=300, widths 3/5/7, V=50 002: the 60 MB
    # word-table gradient goes through GradAllReduce(comm_dtype=bf16) as on the 8-GPU node)
    cfg = synth.DEEPCONN_CFGS[os.environ.get("RBR_TEST_CFG", "cfg1")]

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            m = DeepCoNNpp(cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.0)
        m.load_state_dict(synth.deepconn_params(cfg, 0))
        return m.to(dev).train()

    def batch(seed):
        b = synth.deepconn_batch(cfg, seed)
        return tuple(b[k].to(dev) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")), b["ratings"].to(dev)

    # ---- fp32 data-parallel reference: f32-class conv, exact fp32 all-reduce of every gradient
    m_ref = build()
    sync_ref = GradAllReduce(m_ref)
    o_ref = make_optimizer(m_ref, hip_clip_adam=True)
    # ---- cfg5 variant: bf16 conv class (operands and byte streams), bf16 wire format
    m_b = build()
    sync_b = GradAllReduce(m_b, comm_dtype=torch.bfloat16)
    o_b = make_optimizer(m_b, hip_clip_adam=True)

    # one backward each: the all-reduced gradients inside the bf16 class's tolerance (gradient norms 5e-2, as the 1-GPU test)
    args, ratings = batch(10 + rank)
    RF.set_prod_precision(None)
    torch.nn.functional.mse_loss(m_ref(*args), ratings).backward()
    sync_ref(m_ref)
    RF.set_prod_precision("bf16")
    torch.nn.functional.mse_loss(m_b(*args), ratings).backward()
    sync_b(m_b)
    for (k, pr), pb in zip(m_ref.named_parameters(), m_b.parameters()):
        nr = float(pr.grad.double().norm())
        if nr > 1e-6:
            assert abs(float(pb.grad.double().norm()) - nr) <= 5e-2 * nr, (k, nr, float(pb.grad.double().norm()))
    m_ref.zero_grad(); m_b.zero_grad()

    # three optimisation steps (the bf16 variant replayed as hipGraphs around the eager exchange, as bench.py runs it)
    RF.set_prod_precision("bf16")
    a0, r0 = batch(99)
    stepper = GraphedTrainStep(m_b, o_b, a0, r0, grad_sync=sync_b)
    for step in range(3):
        a2, r2 = batch(20 + 7 * step + rank)
        RF.set_prod_precision(None)
        l_ref, g_ref, _ = train_step(m_ref, o_ref, a2, r2, grad_sync=sync_ref)
        RF.set_prod_precision("bf16")
        l_b, g_b, _ = stepper(a2, r2)
        torch.cuda.synchronize()
        assert abs(float(l_b) - float(l_ref)) <= 2e-2 * max(1.0, abs(float(l_ref))), (step, float(l_b), float(l_ref))
        assert abs(float(g_b) - float(g_ref)) <= 5e-2 * max(1.0, abs(float(g_ref))), (step, float(g_b), float(g_ref))
    RF.set_prod_precision(None)

    # replicas of the bf16 variant hold identical parameters (every rank applies the same averaged gradient)
    for k, p in m_b.named_parameters():
        mine = p.detach().view(torch.int32).to(torch.int64).sum().reshape(1).cpu()
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        assert all(int(v) == int(allv[0]) for v in allv), k
    dist.barrier()
    if rank == 0:
        print("CFG5 BF16 DP OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
