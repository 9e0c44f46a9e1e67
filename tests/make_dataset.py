"""Writes a tiny synthetic dataset in the reference's on-disk format (doc split or review split) for the tests.
The `indexlizer` is an instance of a class from a throw-away module that is removed afterwards, so loading it
exercises the tolerant unpickler exactly as a real meta.pkl (preprocess._tokenizer.Indexlizer) would."""
import importlib
import os
import pickle
import sys
import textwrap

import numpy as np


class _fake_module:
    """Context manager: `with _fake_module(dir, n) as indexlizer:` -- the throw-away module is importable only
    inside the block, so a pickle dumped there references a module that no longer exists afterwards."""

    SRC = textwrap.dedent("""
        def clean_str(s):
            return s
        class Vocab:
            def __init__(self, n):
                self._token2id = {"tok%d" % i: i for i in range(n)}
                self._preprocessor = clean_str
        class Indexlizer:
            def __init__(self, n):
                self._vocab = Vocab(n)
                self._token2id = self._vocab._token2id
                self._tokenizer = clean_str
    """)

    def __init__(self, tmp_dir, vocab_size):
        self.dir = os.path.join(tmp_dir, "_fakepre")
        self.n = vocab_size

    def __enter__(self):
        os.makedirs(self.dir, exist_ok=True)
        with open(os.path.join(self.dir, "fake_tokenizer_mod.py"), "w") as f:
            f.write(self.SRC)
        sys.path.insert(0, self.dir)
        importlib.invalidate_caches()
        return importlib.import_module("fake_tokenizer_mod").Indexlizer(self.n)

    def __exit__(self, *exc):
        sys.path.remove(self.dir)
        sys.modules.pop("fake_tokenizer_mod", None)
        os.remove(os.path.join(self.dir, "fake_tokenizer_mod.py"))
        return False


def write_doc_split(data_dir, n_users=12, n_items=10, vocab=60, doc_len=24, n_train=96, n_valid=32, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(data_dir, exist_ok=True)

    def docs(n):
        out = {}
        for i in range(1, n):
            ln = rng.integers(doc_len // 3, doc_len + 1)
            d = rng.integers(2, vocab, size=doc_len)
            d[ln:] = 0
            out[i] = d.tolist()
        out[0] = [0] * doc_len
        return out

    udocs, idocs = docs(n_users), docs(n_items)
    uq, iq = rng.normal(size=n_users), rng.normal(size=n_items)

    def examples(n):
        ex = []
        for _ in range(n):
            u, i = int(rng.integers(1, n_users)), int(rng.integers(1, n_items))
            r = float(np.clip(np.round(3 + uq[u] + iq[i]), 1, 5))
            ex.append([u, i, r, udocs[u], idocs[i]])
        return ex

    with _fake_module(data_dir, vocab) as indexlizer:
        meta = {"user_num": n_users, "item_num": n_items, "indexlizer": indexlizer, "user_docs": udocs, "item_docs": idocs,
                "doc_len": doc_len}
        with open(os.path.join(data_dir, "meta.pkl"), "wb") as f:
            pickle.dump(meta, f)
    for name, n in (("train", n_train), ("valid", n_valid)):
        with open(os.path.join(data_dir, f"{name}_exmaples.pkl"), "wb") as f:
            pickle.dump(examples(n), f)
    return dict(user_num=n_users, item_num=n_items, vocab=vocab, doc_len=doc_len)


def write_review_split(data_dir, n_users=12, n_items=10, vocab=60, rv_num=4, rv_len=9, n_train=96, n_valid=32, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(data_dir, exist_ok=True)

    def side(n, n_other):
        revs, rids = {}, {}
        for i in range(n):
            k = 0 if i == 0 else int(rng.integers(1, rv_num + 1))
            r = rng.integers(2, vocab, size=(rv_num, rv_len))
            ids = rng.integers(1, n_other, size=rv_num)
            r[k:] = 0
            ids[k:] = 0
            revs[i], rids[i] = r.tolist(), ids.tolist()
        return revs, rids

    urev, urid = side(n_users, n_items)
    irev, irid = side(n_items, n_users)

    def examples(n, train):
        ex = []
        for _ in range(n):
            u, i = int(rng.integers(1, n_users)), int(rng.integers(1, n_items))
            r = float(rng.integers(1, 6))
            row = [u, i, r, urev[u], irev[i], urid[u], irid[i]]
            ex.append(tuple(row + ["dropped"]) if train else tuple(row))
        return ex

    with _fake_module(data_dir, vocab) as indexlizer:
        meta = {"user_num": n_users, "item_num": n_items, "indexlizer": indexlizer, "rv_num": rv_num, "rv_len": rv_len,
                "user_reviews": urev, "item_reviews": irev, "user_rids": urid, "item_rids": irid}
        with open(os.path.join(data_dir, "meta.pkl"), "wb") as f:
            pickle.dump(meta, f)
    for name, n in (("train", n_train), ("valid", n_valid)):
        with open(os.path.join(data_dir, f"{name}_exmaples.pkl"), "wb") as f:
            pickle.dump(examples(n, name == "train"), f)
    return dict(user_num=n_users, item_num=n_items, vocab=vocab, rv_num=rv_num, rv_len=rv_len)


