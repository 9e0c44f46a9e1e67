"""Dataset readers for the reference's on-disk format (CPU only)."""
import os
import pickle

import pytest
import torch

import make_dataset


def test_doc_split_reader_and_tolerant_unpickler(tmp_path):
    from review_based_recommender_amd import data as D
    info = make_dataset.write_doc_split(str(tmp_path / "doc"))
    with pytest.raises(Exception):                       # a plain pickle.load cannot resolve the tokenizer module
        with open(tmp_path / "doc" / "meta.pkl", "rb") as f:
            pickle.load(f)
    ds = D.DocDataset(str(tmp_path / "doc"), "train")
    assert (ds.user_num, ds.item_num, ds.doc_len, ds.vocab_size) == (info["user_num"], info["item_num"], info["doc_len"], info["vocab"])
    assert len(ds) == 96
    batch = ds.collate_fn([ds[i] for i in range(5)])
    u_docs, i_docs, u_m, i_m, u_ids, i_ids, ratings = batch
    assert u_docs.shape == (5, info["doc_len"]) and u_docs.dtype == torch.int64
    assert u_m.dtype == torch.bool and torch.equal(u_m, u_docs != 0)
    assert ratings.dtype == torch.float32 and u_ids.dtype == torch.int64
    ds2 = D.DocDataset(str(tmp_path / "doc"), "valid", with_ids=False)   # D-ATT collate: docs + ratings only
    assert len(ds2.collate_fn([ds2[0], ds2[1]])) == 3


def test_review_split_reader(tmp_path):
    from review_based_recommender_amd import data as D
    info = make_dataset.write_review_split(str(tmp_path / "rev"))
    ds = D.ReviewDataset(str(tmp_path / "rev"), "train")
    assert len(ds[0]) == 7                                # the 8th train field is dropped (train_narre.py:275-277)
    b = ds.collate_fn([ds[i] for i in range(4)])
    assert len(b) == 9
    assert b[0].shape == (4, info["rv_num"], info["rv_len"]) and b[6].shape == (4, info["rv_num"])
    assert torch.equal(b[2], b[0] != 0)


def test_device_cache_gathers_the_same_documents(tmp_path):
    from review_based_recommender_amd import data as D
    make_dataset.write_doc_split(str(tmp_path / "doc"))
    ds = D.DocDataset(str(tmp_path / "doc"), "train")
    cache = D.DeviceDocCache(ds, "cpu")                   # the cache itself is device-agnostic plumbing
    ref = ds.collate_fn([ds[i] for i in range(8)])
    got = cache.doc_batch(ref[4], ref[5])
    for a, b in zip(got, ref[:6]):
        assert torch.equal(a, b)


def test_unpickler_resolves_only_the_allow_list(tmp_path):
    """A crafted pickle that names an importable callable (os.system through REDUCE) loads inert: nothing runs."""
    from review_based_recommender_amd import data as D
    marker = tmp_path / "ran"

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    path = tmp_path / "evil.pkl"
    with open(path, "wb") as f:
        pickle.dump({"user_num": 3, "payload": Evil(), "arr": [1, 2, 3]}, f)
    obj = D.load_pickle(str(path))
    assert not marker.exists(), "the pickle executed a command"
    assert obj["user_num"] == 3 and obj["arr"] == [1, 2, 3]
    assert type(obj["payload"]).__name__ == "system" and isinstance(obj["payload"], D._Placeholder)
    with pytest.raises(RuntimeError):
        obj["payload"]()
    # numpy arrays (what preprocessing may store) still load
    import numpy as np
    with open(path, "wb") as f:
        pickle.dump({"a": np.arange(6, dtype=np.int64).reshape(2, 3), "s": np.float32(2.5)}, f)
    obj = D.load_pickle(str(path))
    assert obj["a"].tolist() == [[0, 1, 2], [3, 4, 5]] and float(obj["s"]) == 2.5


def test_loader_rejects_ids_outside_their_tables(tmp_path):
    """A vocabulary / meta.pkl mismatch is caught at load time, as the IndexError nn.Embedding would raise later."""
    from review_based_recommender_amd import data as D
    info = make_dataset.write_doc_split(str(tmp_path / "doc"))
    path = tmp_path / "doc" / "train_exmaples.pkl"
    ex = D.load_pickle(str(path))
    ex[3][3][0] = info["vocab"]                          # one token id == vocabulary size
    with open(path, "wb") as f:
        pickle.dump(ex, f)
    with pytest.raises(IndexError, match="user document tokens"):
        D.DocDataset(str(tmp_path / "doc"), "train")


def test_data_parallel_loaders_follow_dataparallel_semantics(tmp_path):
    """the DP trainer reshuffles every epoch, splits batch_size over the ranks, and counts every validation
    example exactly once (no sampler padding).  Pure host logic: two 'ranks' built side by side, no process group."""
    from review_based_recommender_amd import data as D
    from review_based_recommender_amd.trainer import DEFAULTS, Args, make_loaders
    make_dataset.write_doc_split(str(tmp_path / "doc"))
    train, valid = D.DocDataset(str(tmp_path / "doc"), "train"), D.DocDataset(str(tmp_path / "doc"), "valid")
    args = Args(dict(DEFAULTS, batch_size=16))
    per_rank = [make_loaders(train, valid, args, r, 2) for r in range(2)]
    assert all(p[4] == 8 for p in per_rank)                               # 16 // 2 examples per rank and step
    orders = {}
    for epoch in (0, 1):
        for r, (_tl, _vl, ts, _vs, _b) in enumerate(per_rank):
            ts.set_epoch(epoch)
            orders[(epoch, r)] = list(iter(ts))
    assert orders[(0, 0)] != orders[(1, 0)] and orders[(0, 1)] != orders[(1, 1)]      # a new permutation every epoch
    for epoch in (0, 1):                                                   # the ranks' shards are disjoint
        assert not set(orders[(epoch, 0)]) & set(orders[(epoch, 1)])
    seen = [i for (_tl, _vl, _ts, vs, _b) in per_rank for i in vs]
    assert sorted(seen) == list(range(len(valid)))                         # every validation example once, none repeated
    n = sum(b[-1].shape[0] for (_tl, vl, _ts, _vs, _b) in per_rank for b in vl)
    assert n == len(valid)
    with pytest.raises(ValueError):
        make_loaders(train, valid, Args(dict(DEFAULTS, batch_size=15)), 0, 2)
    tl, vl, ts, vs, b = make_loaders(train, valid, args, 0, 1)             # one rank: plain shuffled loader, full batch
    assert ts is None and vs is None and b == 16
