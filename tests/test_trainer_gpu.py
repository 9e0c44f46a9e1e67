"""Trainer-equivalent loop on the HIP modules with a tiny synthetic dataset in the reference's format."""
import json
import os
import re

import pytest
import torch

import make_dataset

pytestmark = pytest.mark.gpu

LOG_RE = re.compile(r"^epoch: \d+/\d+, step: \d+/\d+, loss: \d+\.\d{3}, rmse: \d+\.\d{3}, lr: [\d.e-]+, gnorm: \d+\.\d+, time: \d+\.\d{3}$")


def _cfg(tmp_path, kind, data_dir, **extra):
    cfg = {"data_dir": data_dir, "dataset": "synthetic", "log_dir": str(tmp_path / "logs"), "log": True, "log_idx": 2,
           "model_name": kind, "parallel": False, "kernel_sizes": "3,5", "hidden_dim": 8, "embedding_dim": 12, "att_dim": 4,
           "latent_dim": 4, "dropout": 0.5, "arch": "CNN", "use_pretrain": False, "epochs": 3, "batch_size": 16, "lr": 0.002,
           "max_grad_norm": 5.0, "patience": 5, "l_window_size": 5, "l_out_size": 8, "g_out_size": 4, "emb_size": 12,
           "hidden_size_1": 10, "hidden_size_2": 5}
    cfg.update(extra)
    path = tmp_path / f"{kind}.json"
    path.write_text(json.dumps(cfg))
    return str(path)


@pytest.mark.parametrize("kind", ["deepconn", "narre", "dual_att", "simple_siamese"])
def test_trainer_runs_logs_and_checkpoints(tmp_path, kind):
    from review_based_recommender_amd.trainer import ReviewExperiment, parse_args
    data_dir = str(tmp_path / "data")
    (make_dataset.write_review_split if kind in ("narre", "simple_siamese") else make_dataset.write_doc_split)(data_dir)
    exp = ReviewExperiment(kind, parse_args(_cfg(tmp_path, kind, data_dir)), uid="t0")
    exp.train()
    log = open(os.path.join(exp.out_dir, "log.txt")).read().splitlines()
    steps = [l for l in log if l.startswith("epoch:")]
    assert len(steps) == 3 * (6 // 2) and all(LOG_RE.match(l) for l in steps), steps[:2]
    assert sum(l.startswith("valid loss:") for l in log) == 3
    assert any(l.startswith("The total number of trainable parameters") for l in log)
    ck = torch.load(os.path.join(exp.out_dir, "best_model.pt"), map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "updates", "args"}
    assert list(ck["model"].keys()) == list(exp.model.state_dict().keys())
    assert exp.out_dir.endswith(os.path.join("synthetic", kind, "t0"))


def test_early_stop_and_kernel_size_quirk(tmp_path):
    from review_based_recommender_amd.trainer import EarlyStop, ReviewExperiment, parse_args
    data_dir = str(tmp_path / "data")
    make_dataset.write_doc_split(data_dir)
    exp = ReviewExperiment("deepconn", parse_args(_cfg(tmp_path, "deepconn", data_dir, patience=0, hidden_dim=6)),
                           reference_quirks=True, uid="t1")
    assert exp.model.ngram.feature_layer[0].kernel_sizes == [3]        # trainers hard-code [3] (train_deepconn_pp.py:125)
    exp.best_rmse = 0.0                                                  # nothing can improve on it
    with pytest.raises(EarlyStop):
        exp.train()


def test_fast_step_trains_like_the_eager_step(tmp_path):
    """`fast_step: true` (HipClipAdam + hipGraph replay, ragged last batch eager) follows the eager trainer's losses."""
    from review_based_recommender_amd.trainer import ReviewExperiment, parse_args
    data_dir = str(tmp_path / "data")
    make_dataset.write_doc_split(data_dir)
    logs = {}
    for fast in (False, True):
        torch.manual_seed(0)
        exp = ReviewExperiment("deepconn", parse_args(_cfg(tmp_path, "deepconn", data_dir, fast_step=fast, dropout=0.0, batch_size=4)),
                               uid=f"f{int(fast)}")
        exp.train()
        logs[fast] = [l for l in open(os.path.join(exp.out_dir, "log.txt")).read().splitlines() if l.startswith("valid loss:")]
    assert len(logs[True]) == len(logs[False]) == 3
    for a, b in zip(logs[False], logs[True]):
        va, vb = float(a.split()[2].rstrip(",")), float(b.split()[2].rstrip(","))
        assert abs(va - vb) <= 5e-3 * max(1.0, abs(va)), (a, b)


@pytest.mark.parametrize("kind", ["deepconn", "narre", "dual_att"])
def test_one_epoch_follows_the_oracle_step_by_step(tmp_path, kind):
    """f-1 numerics: one epoch with dropout 0 and shuffling off -- every step's training loss and the
    validation RMSE after the epoch against the CPU oracle driven over the SAME batches in the same order
    (trainer/train_deepconn_pp.py:143-168 train loop, :191-232 validation; train_dual_att.py:158-164 for D-ATT)."""
    import math

    from oracle import ref_cpu as O
    from review_based_recommender_amd.trainer import ReviewExperiment, parse_args
    data_dir = str(tmp_path / "data")
    (make_dataset.write_review_split if kind == "narre" else make_dataset.write_doc_split)(data_dir)
    exp = ReviewExperiment(kind, parse_args(_cfg(tmp_path, kind, data_dir, dropout=0.0, shuffle=False, record_steps=True,
                                                 epochs=1, batch_size=16)), uid="o1")
    p0 = {k: v.detach().cpu().clone() for k, v in exp.model.state_dict().items()}
    fwd = {"deepconn": O.deepconn_forward, "narre": (lambda q, *b: O.narre_forward(q, *b)[0]), "dual_att": O.datt_forward}[kind]
    steps = [((lambda q, b=b: fwd(q, *b[:-1])), b[-1]) for b in exp.train_loader]
    exp.train_one_epoch(0)
    exp.valid_one_epoch()
    got = [float(x) for x in exp.step_losses]
    ref_losses, trained = O.train_over_batches(p0, steps, lr=exp.args.lr, max_grad_norm=exp.args.max_grad_norm)
    assert len(got) == len(ref_losses) == len(exp.train_loader) == 6
    for k, (a, b) in enumerate(zip(got, ref_losses)):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (k, a, b)
    sq = n = 0.0
    with torch.no_grad():
        for b in exp.valid_loader:
            pred = fwd(trained, *b[:-1])
            sq += float(((pred - b[-1]) ** 2).sum())
            n += b[-1].numel()
    assert exp.valid_count == len(exp.valid_set) == n
    assert abs(exp.last_valid_rmse - math.sqrt(sq / n)) <= 5e-4 * max(1.0, math.sqrt(sq / n))


def test_reader_and_device_doc_cache_feed_the_model_on_gpu(tmp_path):
    """f-2 on the device: the reader's collate batch and the DeviceDocCache's on-device gather of the same (user, item) ids
    are the same tensors on cuda:0 and give the same predictions.  (The on-disk format itself is parity-UNPINNED: the
    reference holds no dataset fixture; tests/make_dataset.py writes the layout its trainers read.)"""
    from review_based_recommender_amd import data as D
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    from helpers import quiet
    data_dir = str(tmp_path / "data")
    info = make_dataset.write_doc_split(data_dir)
    ds = D.DocDataset(data_dir, "train")
    cache = D.DeviceDocCache(ds, "cuda:0")
    assert cache.user.is_cuda and cache.user.shape == (info["user_num"], info["doc_len"])
    ref = ds.collate_fn([ds[i] for i in range(16)])
    got = cache.doc_batch(ref[4], ref[5])
    for a, b in zip(got, ref[:6]):
        assert a.is_cuda and torch.equal(a.cpu(), b)
    m = quiet(DeepCoNNpp, ds.user_num, ds.item_num, ds.vocab_size, [3, 5], 12, 8, 4, ds.doc_len, None, 0.0).to("cuda:0").eval()
    with torch.no_grad():
        p_cache = m(*got)
        p_host = m(*[t.to("cuda:0") for t in ref[:6]])
    assert torch.equal(p_cache, p_host)
