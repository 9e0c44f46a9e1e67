"""Readers for the reference's on-disk dataset format and a device-resident document cache.

Format (written by the reference's preprocess/divide_and_create_example_{doc,word}.py, read by its trainers):
  <data_dir>/meta.pkl             dict: user_num, item_num, indexlizer, and
      doc split  : user_docs, item_docs (id -> [doc_len] token ids), doc_len     (train_deepconn_pp.py:254-262)
      review split: rv_num, rv_len, user_reviews, item_reviews (id -> [rv_num][rv_len]),
                    user_rids, item_rids (id -> [rv_num] counterpart ids)          (train_narre.py:254-266)
  <data_dir>/{train,valid,test}_exmaples.pkl   (sic)  list of examples
      doc split  : [u_id, i_id, rating, u_doc, i_doc]                              (train_deepconn_pp.py:276)
      review split: 7-tuples (valid/test) or 8-tuples (train; the last field is dropped)  (train_narre.py:273-279)

`meta["indexlizer"]` is an instance of the reference's preprocess._tokenizer.Indexlizer holding function
references (nltk tokenizers, clean_str).  `load_pickle` resolves ONLY an allow-list of globals (plain containers of
builtins / collections and numpy's array reconstructors); every other class or function a pickle names -- importable or
not, `os.system` included -- becomes an inert placeholder: the object graph loads, nothing from it is executed, and only
`_vocab._token2id` (the vocabulary size) is read.

Parity status of the format: UNPINNED.  The reference holds no dataset fixture and its preprocess scripts need nltk, so
the layout above is restated from the trainers' read sites cited here; tests/make_dataset.py writes files of that layout.
The loader also validates every token / user / item id against its table once, at load time (`validate_ranges`).
"""
from __future__ import annotations

import importlib
import os
import pickle
from typing import Sequence

import torch


class _Placeholder:
    """Stands in for a class / function of a module that is not importable in this environment."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    def __call__(self, *a, **k):
        raise RuntimeError("placeholder for an object of a module that is not importable here")


# globals a dataset pickle may legitimately reconstruct: containers and numpy arrays / scalars.  Nothing callable with a
# side effect is on the list; REDUCE on anything else calls a _Placeholder constructor, which does nothing.
_ALLOWED_GLOBALS = {
    "builtins": {"list", "dict", "tuple", "set", "frozenset", "int", "float", "complex", "str", "bytes", "bytearray",
                 "bool", "slice", "range", "object"},
    "collections": {"OrderedDict", "defaultdict", "Counter", "deque"},
    "numpy": {"ndarray", "dtype"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"},
    "numpy._core.multiarray": {"_reconstruct", "scalar"},
    "numpy.core.numeric": {"_frombuffer"},
    "numpy._core.numeric": {"_frombuffer"},
}


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name in _ALLOWED_GLOBALS.get(module, ()):
            return getattr(importlib.import_module(module), name)
        return type(name, (_Placeholder,), {"__module__": module})


def load_pickle(path):
    with open(path, "rb") as f:
        return _TolerantUnpickler(f).load()


def vocab_size_of(indexlizer) -> int:
    """len(indexlizer._vocab)  (preprocess/_tokenizer.py:73-74: the size of _token2id)."""
    vocab = getattr(indexlizer, "_vocab", None)
    for holder in (vocab, indexlizer):
        t2i = getattr(holder, "_token2id", None) if holder is not None else None
        if t2i is not None:
            return len(t2i)
    if isinstance(indexlizer, int):
        return int(indexlizer)
    raise ValueError("cannot find the vocabulary (_vocab._token2id) in meta['indexlizer']")


def _check_range(values, limit: int, what: str) -> None:
    t = torch.as_tensor(values, dtype=torch.int64)
    if t.numel() and (int(t.min()) < 0 or int(t.max()) >= limit):
        raise IndexError(f"{what}: ids span [{int(t.min())}, {int(t.max())}] but the table has {limit} rows "
                         "(vocabulary / meta.pkl mismatch?)")


def get_mask(tensor: torch.Tensor, padding_idx: int = 0) -> torch.Tensor:
    """utils.py:30-42 -- bool mask, False where the token equals padding_idx."""
    return tensor != padding_idx


def _rows(mapping, n, what):
    """id -> row table as a list of length n (dicts keyed by id or sequences indexed by id)."""
    if isinstance(mapping, dict):
        any_row = next(iter(mapping.values()))
        zero = [[0] * len(any_row[0]) for _ in any_row] if isinstance(any_row[0], (list, tuple)) else [0] * len(any_row)
        return [mapping.get(i, zero) for i in range(n)]
    if len(mapping) < n:
        raise ValueError(f"{what} has {len(mapping)} rows for {n} ids")
    return list(mapping[:n])


class DocDataset(torch.utils.data.Dataset):
    """doc split (DeepCoNN / D-ATT): examples are [u_id, i_id, rating, u_doc, i_doc]."""

    def __init__(self, data_dir: str, set_name: str, with_ids: bool = True):
        meta = load_pickle(os.path.join(data_dir, "meta.pkl"))
        self.user_num, self.item_num = meta["user_num"], meta["item_num"]
        self.doc_len = meta["doc_len"]
        self.vocab_size = vocab_size_of(meta["indexlizer"])
        self.user_docs, self.item_docs = meta["user_docs"], meta["item_docs"]
        self.examples = load_pickle(os.path.join(data_dir, f"{set_name}_exmaples.pkl"))
        self.with_ids = with_ids
        self.validate_ranges()

    def validate_ranges(self) -> None:
        """Every token / user / item id of the split against its table, once at load time: the IndexError nn.Embedding
        would raise on the first bad batch (models/deepconn/layers.py:23), up front.  A model fed from a validated dataset
        may run with validate_ids = False."""
        _check_range([e[0] for e in self.examples], self.user_num, "user ids")
        _check_range([e[1] for e in self.examples], self.item_num, "item ids")
        _check_range([e[3] for e in self.examples], self.vocab_size, "user document tokens")
        _check_range([e[4] for e in self.examples], self.vocab_size, "item document tokens")

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        return self.examples[i][:5]

    def collate_fn(self, batch):
        """train_deepconn_pp.py:281-292 (with ids) / train_dual_att.py:273-280 (docs and ratings only)."""
        u_ids, i_ids, ratings, u_docs, i_docs = zip(*batch)
        u_docs, i_docs = torch.LongTensor(u_docs), torch.LongTensor(i_docs)
        ratings = torch.FloatTensor(ratings)
        if not self.with_ids:
            return u_docs, i_docs, ratings
        return (u_docs, i_docs, get_mask(u_docs), get_mask(i_docs), torch.LongTensor(u_ids), torch.LongTensor(i_ids), ratings)


class ReviewDataset(torch.utils.data.Dataset):
    """review split (NARRE): examples are (u_id, i_id, rating, u_revs, i_revs, u_rids, i_rids[, extra])."""

    def __init__(self, data_dir: str, set_name: str):
        meta = load_pickle(os.path.join(data_dir, "meta.pkl"))
        self.user_num, self.item_num = meta["user_num"], meta["item_num"]
        self.rv_num, self.rv_len = meta["rv_num"], meta["rv_len"]
        self.vocab_size = vocab_size_of(meta["indexlizer"])
        self.user_reviews, self.item_reviews = meta["user_reviews"], meta["item_reviews"]
        self.user_rids, self.item_rids = meta["user_rids"], meta["item_rids"]
        self.examples = load_pickle(os.path.join(data_dir, f"{set_name}_exmaples.pkl"))
        self.validate_ranges()

    def validate_ranges(self) -> None:
        _check_range([e[0] for e in self.examples], self.user_num, "user ids")
        _check_range([e[1] for e in self.examples], self.item_num, "item ids")
        _check_range([e[3] for e in self.examples], self.vocab_size, "user review tokens")
        _check_range([e[4] for e in self.examples], self.vocab_size, "item review tokens")
        _check_range([e[5] for e in self.examples], self.item_num, "user-side counterpart (item) ids")
        _check_range([e[6] for e in self.examples], self.user_num, "item-side counterpart (user) ids")

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        return self.examples[i][:7]     # train examples carry an 8th field that the trainer drops

    def collate_fn(self, batch):
        """train_narre.py:316-330."""
        u_ids, i_ids, ratings, u_revs, i_revs, u_rids, i_rids = zip(*batch)
        u_revs, i_revs = torch.LongTensor(u_revs), torch.LongTensor(i_revs)
        return (u_revs, i_revs, get_mask(u_revs), get_mask(i_revs), torch.LongTensor(u_ids), torch.LongTensor(i_ids),
                torch.LongTensor(u_rids), torch.LongTensor(i_rids), torch.FloatTensor(ratings))


class DeviceDocCache:
    """All user / item documents resident on the GPU, so a batch is a pair of id vectors gathered ON DEVICE
    instead of 2 x doc_len token ids per pair shipped from the DataLoader workers.

    doc split: user_docs [U, L], item_docs [I, L];  review split: reviews [U, R, T] plus counterpart ids [U, R].
    The reference trains on the examples' own copies (train_deepconn_pp.py:276); for the doc split those are
    the same per-id documents, so gathering by id is equivalent.  (For NARRE's train split the reference blanks
    the target review inside each example, divide_and_create_example_word.py:262-288 -- use the examples there.)"""

    def __init__(self, ds, device):
        self.device = torch.device(device)
        if isinstance(ds, ReviewDataset):
            self.user = torch.tensor(_rows(ds.user_reviews, ds.user_num, "user_reviews"), dtype=torch.int64, device=self.device)
            self.item = torch.tensor(_rows(ds.item_reviews, ds.item_num, "item_reviews"), dtype=torch.int64, device=self.device)
            self.user_rids = torch.tensor(_rows(ds.user_rids, ds.user_num, "user_rids"), dtype=torch.int64, device=self.device)
            self.item_rids = torch.tensor(_rows(ds.item_rids, ds.item_num, "item_rids"), dtype=torch.int64, device=self.device)
        else:
            self.user = torch.tensor(_rows(ds.user_docs, ds.user_num, "user_docs"), dtype=torch.int64, device=self.device)
            self.item = torch.tensor(_rows(ds.item_docs, ds.item_num, "item_docs"), dtype=torch.int64, device=self.device)
            self.user_rids = self.item_rids = None

    def doc_batch(self, u_ids: torch.Tensor, i_ids: torch.Tensor):
        u_ids, i_ids = u_ids.to(self.device), i_ids.to(self.device)
        u_docs, i_docs = self.user.index_select(0, u_ids), self.item.index_select(0, i_ids)
        return u_docs, i_docs, get_mask(u_docs), get_mask(i_docs), u_ids, i_ids

    def review_batch(self, u_ids: torch.Tensor, i_ids: torch.Tensor):
        u_ids, i_ids = u_ids.to(self.device), i_ids.to(self.device)
        u, i = self.user.index_select(0, u_ids), self.item.index_select(0, i_ids)
        return (u, i, get_mask(u), get_mask(i), u_ids, i_ids, self.user_rids.index_select(0, u_ids),
                self.item_rids.index_select(0, i_ids))
