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
