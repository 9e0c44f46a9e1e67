"""Data-parallel path on CPU: 2 processes, gloo backend, 127.0.0.1 rendezvous.

Checks what the RCCL path must guarantee by construction: after GradAllReduce the
gradients of every rank equal the single-process full-batch gradient (equal shards: mean of shard
means == global mean), parameters stay identical across ranks after the optimiser step, and
clip_grad_norm_ sees the global gradient.  The model here is the CPU oracle's DeepCoNN wrapped in an
nn.Module -- the HIP kernels cannot run without a GPU, and the hook under test is device-agnostic."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleDeepCoNN(nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.keys = list(sd.keys())
        self.params = nn.ParameterList([nn.Parameter(v.clone()) for v in sd.values()])

    def forward(self, *args):
        from oracle import ref_cpu as O
        return O.deepconn_forward(dict(zip(self.keys, self.params)), *args)


def _worker(rank, world, port, out_dir, comm_dtype):
    for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import synth
    from review_based_recommender_amd.distributed import (GradAllReduce, broadcast_parameters,
                                                          init_process_group_from_env, shard_batch)
    from review_based_recommender_amd.train_step import make_optimizer, train_step
    torch.set_num_threads(2)
    init_process_group_from_env("gloo")
    cfg = dict(synth.DEEPCONN_CFGS["small"])
    cfg["V"] = 40000     # word table of 960k elements... below the big-bucket threshold; see test_big_bucket below
    sd = synth.deepconn_params(cfg, 0)
    if rank != 0:        # deliberately different initial weights: broadcast must fix them
        sd = {k: v + 1.0 for k, v in sd.items()}
    model = _OracleDeepCoNN(sd)
    broadcast_parameters(model)
    b = synth.deepconn_batch(cfg, 1)
    full = (b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"], b["ratings"])
    shard = shard_batch(full, rank, world)
    sync = GradAllReduce(model, comm_dtype=comm_dtype)
    opt = make_optimizer(model)
    loss, gnorm, _ = train_step(model, opt, shard[:-1], shard[-1], grad_sync=sync)
    grads = {k: p.grad.clone() for k, p in zip(model.keys, model.params)}
    params = {k: p.detach().clone() for k, p in zip(model.keys, model.params)}
    torch.save(dict(grads=grads, params=params, gnorm=gnorm, loss=loss), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm_dtype", [None, torch.bfloat16])
def test_two_rank_dp_equals_full_batch(tmp_path, comm_dtype):
    import synth
    from review_based_recommender_amd import distributed as D
    from review_based_recommender_amd.train_step import make_optimizer, train_step
    old = D.BIG_BUCKET_ELEMS
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), comm_dtype), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")

    cfg = dict(synth.DEEPCONN_CFGS["small"])
    cfg["V"] = 40000
    sd = synth.deepconn_params(cfg, 0)
    model = _OracleDeepCoNN(sd)
    b = synth.deepconn_batch(cfg, 1)
    opt = make_optimizer(model)
    args = (b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"])
    loss = torch.nn.functional.mse_loss(model(*args), b["ratings"])
    loss.backward()
    ref = {k: p.grad.clone() for k, p in zip(model.keys, model.params)}
    gnorm_ref = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)

    exact = comm_dtype is None
    for k in ref:
        # DP grads were clipped in place after the sync: undo with the recorded norm
        scale = max(1.0, float(r0["gnorm"]) / 5.0)
        g0, g1 = r0["grads"][k] * scale, r1["grads"][k] * scale
        assert torch.equal(r0["grads"][k], r1["grads"][k]), f"ranks disagree on grad {k}"
        tol = 1e-5 if exact or ref[k].numel() < D.BIG_BUCKET_ELEMS else 1e-2
        denom = float(ref[k].norm()) + 1e-12
        assert float((g0 - ref[k]).norm()) / denom <= tol, k
        assert torch.equal(r0["params"][k], r1["params"][k]), f"replicas diverged on {k}"
    if exact:
        assert abs(float(r0["gnorm"]) - float(gnorm_ref)) <= 1e-5 * float(gnorm_ref)
    assert D.BIG_BUCKET_ELEMS == old


def _bucket_worker(rank, world, port, out_dir):
    for p in (ROOT,):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from review_based_recommender_amd import distributed as D
    D.init_process_group_from_env("gloo")
    D.BIG_BUCKET_ELEMS = 1000    # force the in-place big-tensor path on a small model
    m = nn.Sequential(nn.Linear(50, 40), nn.Linear(40, 3))     # 2000-element weight -> "big", rest "small"
    for i, p in enumerate(m.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    D.GradAllReduce(m)(m)
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(m.parameters()))
    torch.save(ok, os.path.join(out_dir, f"ok{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_big_and_small_buckets_average(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert torch.load(tmp_path / "ok0.pt") and torch.load(tmp_path / "ok1.pt")


def test_shard_batch_is_contiguous_equal_split():
    from review_based_recommender_amd.distributed import shard_batch
    x = torch.arange(12).view(6, 2)
    a, = shard_batch((x,), 1, 3)
    assert a.tolist() == [[4, 5], [6, 7]]
    with pytest.raises(AssertionError):
        shard_batch((x,), 0, 4)
