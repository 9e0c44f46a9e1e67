"""Worker of tests/test_tap_exchange_gpu.py: launched by torch.distributed.run, 2+ ranks on ONE GPU over gloo.
Checks the tap exchange (distributed.TapExchange) against the dense all-reduce of the same gradients."""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch
import torch.distributed as dist
import torch.nn.functional as F

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
    _lib.lib().rbr_set_conv_mode(2)                       # token-product path also at this small shape
    cfg = synth.DEEPCONN_CFGS["small"]

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            m = DeepCoNNpp(cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.0)
        m.load_state_dict(synth.deepconn_params(cfg, 0))
        return m.to(dev).train()

    def batch(seed):
        b = synth.deepconn_batch(cfg, seed)
        return tuple(b[k].to(dev) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")), b["ratings"].to(dev)

    args, ratings = batch(10 + rank)                       # every rank its own shard
    owner = os.environ.get("RBR_TEST_EXCHANGE", "taps") == "owner"      # the owner-partitioned rebuild + slab all-gather

    # dense all-reduce of every gradient
    m_d = build()
    sync_d = GradAllReduce(m_d)
    F.mse_loss(m_d(*args), ratings).backward()
    sync_d(m_d)
    ref = {k: p.grad.clone() for k, p in m_d.named_parameters()}

    # tap exchange of the table gradient
    m_t = build()
    o_t = make_optimizer(m_t, hip_clip_adam=True)
    sync_t = GradAllReduce(m_t, tap_table=m_t.word_embeddings.embedding.weight, owner=owner, optimizer=o_t if owner else None)
    F.mse_loss(m_t(*args), ratings).backward()
    assert m_t.word_embeddings.embedding.weight.grad is None, "the table gradient must come from the exchange"
    sync_t(m_t)
    if owner:
        assert m_t.word_embeddings.embedding.weight.grad is None and len(o_t._row_grads) == 1, "row form expected"
        o_t.materialize_grads()                            # the dense view of the exchanged rows, for the comparison below
        sync_t.tap.check()
    for k, p in m_t.named_parameters():
        scale = float(ref[k].abs().max()) + 1e-12
        err = float((p.grad - ref[k]).abs().max())
        assert err <= 1e-6 + 2e-5 * scale, (k, err, scale)

    # replicas hold bit-identical table gradients
    g = m_t.word_embeddings.embedding.weight.grad
    mine = g.view(torch.int32).to(torch.int64).sum().reshape(1).cpu()
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    assert all(int(v) == int(allv[0]) for v in allv), [int(v) for v in allv]

    # three optimisation steps, eager and replayed as hipGraphs, against the dense data-parallel step
    o_d = make_optimizer(m_d, hip_clip_adam=True)
    m_d.zero_grad(); o_t.zero_grad()
    stepper = GraphedTrainStep(m_t, o_t, args, ratings, grad_sync=sync_t)
    for step in range(3):
        a2, r2 = batch(20 + 7 * step + rank)
        ld, gd, _ = train_step(m_d, o_d, a2, r2, grad_sync=sync_d)
        lt, gt, _ = stepper(a2, r2)
        torch.cuda.synchronize()
        # step 0 sees identical parameters; afterwards Adam has turned rounding-level gradient differences (summation
        # order of the table gradient) into +-lr steps on near-zero gradients, as in the single-GPU parity tests
        tol = 1e-5 if step == 0 else 2e-4
        assert abs(float(ld) - float(lt)) <= tol * max(1.0, abs(float(ld))), (step, float(ld), float(lt))
        assert abs(float(gd) - float(gt)) <= max(tol, 1e-4) * max(1.0, abs(float(gd))), (step, float(gd), float(gt))
    # an eager step between replays (a trainer's ragged last batch) takes the same exchange and stays correct
    a3, r3 = batch(40 + rank)
    ld, _, _ = train_step(m_d, o_d, a3, r3, grad_sync=sync_d)
    lt, _, _ = train_step(m_t, o_t, a3, r3, grad_sync=sync_t)
    a4, r4 = batch(50 + rank)
    train_step(m_d, o_d, a4, r4, grad_sync=sync_d)
    stepper(a4, r4)
    torch.cuda.synchronize()
    assert abs(float(ld) - float(lt)) <= 2e-4 * max(1.0, abs(float(ld))), (float(ld), float(lt))
    # replicas of the tap-exchange model are still bit-identical after the steps (every parameter)
    chk = torch.stack([p.detach().view(torch.int32).to(torch.int64).sum() for p in m_t.parameters()]).cpu()
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    assert all(torch.equal(c, allc[0]) for c in allc), "replicas diverged"
    # and close to the dense data-parallel model (same bounds as the single-GPU parity tests)
    worst = (0.0, 0.0, "")
    for (k, pd), pt in zip(m_d.named_parameters(), m_t.parameters()):
        d = (pd.detach() - pt.detach()).abs()
        worst = max(worst, (float(d.max()), float(d.pow(2).mean().sqrt()), k))
        assert float(d.max()) <= 1e-3 and float(d.pow(2).mean().sqrt()) <= 1e-4, (k, float(d.max()))
    if rank == 0:
        print("largest parameter difference after 3 steps (max, rms, name):", worst, flush=True)
    if owner:
        sync_t.tap.check()
        assert m_t.word_embeddings.embedding.weight.grad is None, "owner mode: the optimizer reads the exchanged rows in place"
        # an owner whose share of the taps overflows its sort must not leave a wrong gradient behind, and no
        # rank may hang.  ONE rank raises the flag here (two ranks cannot overflow for real: the sort is sized for twice the even
        # share; the kernel-side flag is tested in test_rebuild_from_taps_matches_float64_reference_and_is_order_free): the flag
        # travels with the slab, BOTH ranks rebuild the step the replicated way, and the step equals the dense data-parallel one.
        assert sync_t.tap.overflow_fallbacks == 0
        sync_t.tap._test_force_overflow = (rank == 1)
        with torch.no_grad():
            for pd, pt in zip(m_d.parameters(), m_t.parameters()):
                pt.copy_(pd)
        a5, r5 = batch(60 + rank)
        m_d.zero_grad(); o_t.zero_grad()
        F.mse_loss(m_d(*a5), r5).backward()
        sync_d(m_d)
        F.mse_loss(m_t(*a5), r5).backward()
        sync_t(m_t)
        torch.cuda.synchronize()
        assert sync_t.tap.overflow_fallbacks == 1, "every rank must take the fallback when any rank's flag is up"
        gt = m_t.word_embeddings.embedding.weight.grad
        assert gt is not None and len(o_t._row_grads) == 0, "the fallback hands over a dense gradient"
        gd = m_d.word_embeddings.embedding.weight.grad
        assert float((gt - gd).abs().max()) <= 1e-6 + 2e-5 * float(gd.abs().max())
        o_t.clip_and_step(5.0)                                 # and the optimizer takes it (dense path)
        sync_t.tap._test_force_overflow = False
    RF.set_tap_sink(None)
    dist.barrier()
    if rank == 0:
        print("TAP EXCHANGE OK" + (" (owner)" if owner else ""), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
