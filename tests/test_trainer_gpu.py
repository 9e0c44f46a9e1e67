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
