"""Seeded synthetic parameters and review batches.

Own code, shared by three users so that they all see bit-identical tensors:
  * tests/golden/make_golden.py  (runs the *reference* on them, in the build container only)
  * tests/                       (run the oracle and the HIP path on them)
  * bench.py                     (workload generator)

Everything is drawn from numpy's PCG64 (`np.random.default_rng`), whose stream is
stable across numpy versions and platforms, then wrapped as torch CPU tensors.
Parameter dictionaries use the reference's `state_dict` key names
(SURVEY.md §5 "checkpoint / resume") so one dict loads into the reference modules
(via load_state_dict) and into ours.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


# --------------------------------------------------------------------------- configs
# Named shapes. cfg1..cfg4 follow BASELINE.json `configs` / SURVEY.md §8 header.
DEEPCONN_CFGS = {
    # tiny: exercises two widths, all-pad doc, id 0, non-prefix mask
    "tiny": dict(B=4, L=16, D=8, kz=[3, 5], H=6, K=4, V=20, U=5, I=5),
    # small: three widths, L not a multiple of 32, D not a multiple of 20
    "small": dict(B=8, L=50, D=24, kz=[3, 5, 7], H=30, K=8, V=200, U=30, I=30),
    # single odd width, D=100 (cfg1-like but tiny batch) -- the trainer's hard-coded [3]
    "k3": dict(B=6, L=70, D=100, kz=[3], H=150, K=32, V=500, U=50, I=50),
    "cfg1": dict(B=32, L=300, D=100, kz=[3], H=150, K=32, V=8000, U=1001, I=1001),
    "cfg2": dict(B=256, L=512, D=300, kz=[3, 5, 7], H=150, K=32, V=50002, U=1001, I=1001),
}

NARRE_CFGS = {
    "tiny": dict(B=3, R=4, T=8, D=8, kz=[3], H=6, A=4, K=4, V=20, U=5, I=5),
    "small": dict(B=5, R=6, T=21, D=20, kz=[3], H=12, A=8, K=8, V=100, U=20, I=20),
    "cfg3": dict(B=256, R=10, T=50, D=300, kz=[3], H=150, A=32, K=32, V=50002, U=1001, I=1001),
}

# SimpleSiamese (models/simple_siamese): bag-of-embeddings reviews + additive attention over reviews.
# LT: latent_transform (Linear D->K + Tanh per review), UB: use_ui_bias (FM vs FMWithoutUIBias)
SIAMESE_CFGS = {
    "tiny": dict(B=3, R=4, T=8, D=8, K=4, V=20, U=5, I=5, LT=False, UB=True),
    "small": dict(B=5, R=6, T=21, D=20, K=8, V=100, U=20, I=20, LT=True, UB=False),
    "toys": dict(B=64, R=11, T=50, D=108, K=32, V=50002, U=1001, I=1001, LT=False, UB=True),   # defalut_simple_train.json
}

DATT_CFGS = {
    "tiny": dict(B=2, L=16, E=6, win=5, l_out=8, g_out=4, h1=10, h2=5, V=20),
    "small": dict(B=4, L=40, E=20, win=5, l_out=24, g_out=12, h1=32, h2=8, V=100),
    "cfg4": dict(B=512, L=1024, E=100, win=5, l_out=200, g_out=100, h1=500, h2=50, V=50002),
}


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _uniform(rng, shape, bound):
    return _t(rng.uniform(-bound, bound, size=shape).astype(np.float32))


# --------------------------------------------------------------------------- params
def _word_table(rng, V, D, scale=1.0):
    w = (rng.standard_normal((V, D)) * scale).astype(np.float32)
    w[0] = 0.0  # nn.Embedding(padding_idx=0) zeroes the pad row at init
    return _t(w)


def _conv_params(rng, sd, prefix, kz_list, D, H):
    per = H // len(kz_list)
    for i, kz in enumerate(kz_list):
        bound = 1.0 / math.sqrt(D * kz)  # nn.Conv1d default init range
        sd[f"{prefix}.{i}.weight"] = _uniform(rng, (per, D, kz), bound)
        sd[f"{prefix}.{i}.bias"] = _uniform(rng, (per,), bound)


def _lastfeat(rng, sd, name, n_ids, H, K):
    sd[f"{name}.W"] = _uniform(rng, (H, K), 0.1)
    sd[f"{name}.b"] = torch.full((K,), 0.1)
    sd[f"{name}.ebd.weight"] = _uniform(rng, (n_ids, K), 0.1)  # pad row NOT zero (quirk 4)


def _fm(rng, sd, U, I, K):
    sd["fm.h"] = _uniform(rng, (K, 1), 0.1)
    sd["fm.g_bias"] = torch.full((1,), 0.1)
    sd["fm.user_bias.weight"] = _uniform(rng, (U, 1), 0.1)
    sd["fm.item_bias.weight"] = _uniform(rng, (I, 1), 0.1)


def deepconn_params(cfg, seed=0, table_scale=1.0):
    rng = np.random.default_rng(seed)
    c = cfg
    sd = OrderedDict()
    sd["word_embeddings.embedding.weight"] = _word_table(rng, c["V"], c["D"], table_scale)
    _conv_params(rng, sd, "ngram.feature_layer.0.list_of_conv1d", c["kz"], c["D"], c["H"])
    _lastfeat(rng, sd, "user_feat", c["U"], c["H"], c["K"])
    _lastfeat(rng, sd, "item_feat", c["I"], c["H"], c["K"])
    _fm(rng, sd, c["U"], c["I"], c["K"])
    return sd


def deepconn_hier_params(cfg, seed=0):
    """arch="HierPooling": no conv; optional Linear(D->H) proj when D != H."""
    rng = np.random.default_rng(seed)
    c = cfg
    sd = OrderedDict()
    sd["word_embeddings.embedding.weight"] = _word_table(rng, c["V"], c["D"])
    if c["D"] != c["H"]:
        bound = 1.0 / math.sqrt(c["D"])
        sd["ngram.feature_layer.0.proj_layer.weight"] = _uniform(rng, (c["H"], c["D"]), bound)
        sd["ngram.feature_layer.0.proj_layer.bias"] = _uniform(rng, (c["H"],), bound)
    _lastfeat(rng, sd, "user_feat", c["U"], c["H"], c["K"])
    _lastfeat(rng, sd, "item_feat", c["I"], c["H"], c["K"])
    _fm(rng, sd, c["U"], c["I"], c["K"])
    return sd


def _linatt(rng, sd, name, n_ids, H, A):
    sd[f"{name}.W_rv"] = _uniform(rng, (H, A), 0.1)
    sd[f"{name}.W_id"] = _uniform(rng, (A, A), 0.1)
    sd[f"{name}.h"] = _uniform(rng, (A, 1), 0.1)
    sd[f"{name}.b_1"] = torch.full((A,), 0.1)
    sd[f"{name}.b_2"] = torch.full((1,), 0.1)
    e = rng.standard_normal((n_ids, A)).astype(np.float32)
    e[0] = 0.0  # nn.Embedding(padding_idx) default init, never re-initialised in narre.py:36
    sd[f"{name}.ebd_vals.weight"] = _t(e)


def narre_params(cfg, seed=0):
    rng = np.random.default_rng(seed)
    c = cfg
    sd = OrderedDict()
    sd["word_embeddings.embedding.weight"] = _word_table(rng, c["V"], c["D"])
    _conv_params(rng, sd, "ngram.feature_layer.0.list_of_conv1d", c["kz"], c["D"], c["H"])
    _linatt(rng, sd, "user_att", c["I"], c["H"], c["A"])  # user tower keyed by ITEM ids
    _linatt(rng, sd, "item_att", c["U"], c["H"], c["A"])
    _lastfeat(rng, sd, "user_feat", c["U"], c["H"], c["K"])
    _lastfeat(rng, sd, "item_feat", c["I"], c["H"], c["K"])
    _fm(rng, sd, c["U"], c["I"], c["K"])
    return sd


def siamese_params(cfg, seed=0):
    """state_dict of models/simple_siamese/simple_siamese.py:8-36 (reset_parameters bounds of its layers.py)."""
    rng = np.random.default_rng(seed)
    c = cfg
    D, K = c["D"], c["K"]
    H = K if c["LT"] else D                  # review feature width after the optional latent transform
    sd = OrderedDict()
    sd["word_embedding.embedding.weight"] = _word_table(rng, c["V"], D)
    if c["LT"]:
        b = 1.0 / math.sqrt(D)
        sd["latent_transform_layer.0.weight"] = _uniform(rng, (K, D), b)
        sd["latent_transform_layer.0.bias"] = _uniform(rng, (K,), b)
    _lastfeat(rng, sd, "user_last_feat_layer", c["U"], H, K)
    _lastfeat(rng, sd, "item_last_feat_layer", c["I"], H, K)
    b = 1.0 / math.sqrt(H)
    sd["review_att_layer.proj_layer.0.weight"] = _uniform(rng, (K, H), b)
    sd["review_att_layer.proj_layer.0.bias"] = _uniform(rng, (K,), b)
    sd["review_att_layer.inner_product.weight"] = _uniform(rng, (1, K), 1.0 / math.sqrt(K))
    sd["fm.h"] = _uniform(rng, (K, 1), 0.1)
    sd["fm.g_bias"] = torch.full((1,), 0.1)          # state_dict order: a module's own parameters, then its children
    if c["UB"]:
        sd["fm.user_bias.weight"] = _uniform(rng, (c["U"], 1), 0.1)
        sd["fm.item_bias.weight"] = _uniform(rng, (c["I"], 1), 0.1)
    return sd


def datt_params(cfg, seed=0, table_scale=1.0):
    rng = np.random.default_rng(seed)
    c = cfg
    E, L = c["E"], c["L"]
    sd = OrderedDict()
    sd["word_embeddings.embedding.weight"] = _word_table(rng, c["V"], E, table_scale)
    for side in ("u", "i"):
        p = f"{side}_local_atten"
        b = 1.0 / math.sqrt(E * c["win"])
        sd[f"{p}.attn.0.weight"] = _uniform(rng, (1, E, c["win"]), b)
        sd[f"{p}.attn.0.bias"] = _uniform(rng, (1,), b)
        b = 1.0 / math.sqrt(E)
        sd[f"{p}.conv.0.weight"] = _uniform(rng, (c["l_out"], E, 1), b)
        sd[f"{p}.conv.0.bias"] = _uniform(rng, (c["l_out"],), b)
        p = f"{side}_global_atten"
        b = 1.0 / math.sqrt(E * L)
        sd[f"{p}.attn.0.weight"] = _uniform(rng, (1, E, L), b)
        sd[f"{p}.attn.0.bias"] = _uniform(rng, (1,), b)
        for n, k in (("conv1", 2), ("conv2", 3), ("conv3", 4)):
            b = 1.0 / math.sqrt(E * k)
            sd[f"{p}.{n}.0.weight"] = _uniform(rng, (c["g_out"], E, k), b)
            sd[f"{p}.{n}.0.bias"] = _uniform(rng, (c["g_out"],), b)
    fc_in = c["l_out"] + 3 * c["g_out"]
    b = 1.0 / math.sqrt(fc_in)
    sd["fc.0.weight"] = _uniform(rng, (c["h1"], fc_in), b)
    sd["fc.0.bias"] = _uniform(rng, (c["h1"],), b)
    b = 1.0 / math.sqrt(c["h1"])
    sd["fc.3.weight"] = _uniform(rng, (c["h2"], c["h1"]), b)
    sd["fc.3.bias"] = _uniform(rng, (c["h2"],), b)
    return sd


# --------------------------------------------------------------------------- inputs
def _zipf_ids(rng, shape, V, s=1.07):
    """Zipf(s) over token ids [2, V) (0 = pad, 1 = unk are never drawn)."""
    n = V - 2
    ranks = np.arange(1, n + 1, dtype=np.float64)
    p = ranks ** (-s)
    cdf = np.cumsum(p / p.sum())
    u = rng.random(size=shape)
    return (np.searchsorted(cdf, u, side="left").clip(0, n - 1) + 2).astype(np.int64)


def _docs(rng, n, L, V, min_frac=0.25):
    """n right-padded docs: length ~ U[ceil(L*min_frac), L], pad id 0 (prefix masks)."""
    ids = _zipf_ids(rng, (n, L), V)
    lo = max(1, int(math.ceil(L * min_frac)))
    lens = rng.integers(lo, L + 1, size=n)
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    return ids


def deepconn_batch(cfg, seed=1, edge_cases=False):
    """Returns dict of CPU tensors: u_docs,i_docs [B,L] int64; masks bool; ids int64; ratings f32."""
    rng = np.random.default_rng(seed)
    c = cfg
    B, L, V = c["B"], c["L"], c["V"]
    u_docs = _docs(rng, B, L, V)
    i_docs = _docs(rng, B, L, V)
    u_ids = rng.integers(1, c["U"], size=B).astype(np.int64)
    i_ids = rng.integers(1, c["I"], size=B).astype(np.int64)
    ratings = rng.integers(1, 6, size=B).astype(np.float32)
    u_mask = u_docs != 0
    i_mask = i_docs != 0
    if edge_cases:
        u_docs[0, :] = 0          # all-pad document -> feature = relu(conv bias)
        u_mask[0, :] = False
        u_ids[0] = 0              # id 0: LastFeat/FM pad rows are non-zero (quirk 4)
        if B > 1:
            i_ids[1] = 0
            # non-prefix mask: arbitrary holes although the token ids are non-zero
            i_mask[1, ::3] = False
            # mask True on a pad token (row 0 of the table is used as-is)
            i_docs[1, 1] = 0
            i_mask[1, 1] = True
        if B > 2:
            # duplicated n-grams -> exact ties in the max-pool (first index must win)
            half = L // 2
            u_docs[2, half:2 * half] = u_docs[2, :half]
            u_mask[2] = u_docs[2] != 0
    return dict(
        u_docs=_t(u_docs, torch.int64), i_docs=_t(i_docs, torch.int64),
        u_masks=_t(u_mask, torch.bool), i_masks=_t(i_mask, torch.bool),
        u_ids=_t(u_ids, torch.int64), i_ids=_t(i_ids, torch.int64),
        ratings=_t(ratings),
    )


def narre_batch(cfg, seed=1, edge_cases=False):
    rng = np.random.default_rng(seed)
    c = cfg
    B, R, T, V = c["B"], c["R"], c["T"], c["V"]

    def side(n_other):
        txt = _docs(rng, B * R, T, V).reshape(B, R, T)
        rid = rng.integers(1, n_other, size=(B, R)).astype(np.int64)
        nrev = rng.integers(1, R + 1, size=B)
        dead = np.arange(R)[None, :] >= nrev[:, None]     # trailing reviews: all pad, id 0
        txt[dead] = 0
        rid[dead] = 0
        return txt, rid

    u_text, reuid = side(c["I"])   # user's reviews are keyed by the item they were written for
    i_text, reiid = side(c["U"])
    u_id = rng.integers(1, c["U"], size=B).astype(np.int64)
    i_id = rng.integers(1, c["I"], size=B).astype(np.int64)
    ratings = rng.integers(1, 6, size=B).astype(np.float32)
    if edge_cases:
        u_text[0] = 0
        reuid[0] = 0               # a user with no review at all
        u_id[0] = 0
    return dict(
        u_text=_t(u_text, torch.int64), i_text=_t(i_text, torch.int64),
        u_masks=_t(u_text != 0, torch.bool), i_masks=_t(i_text != 0, torch.bool),
        u_id=_t(u_id, torch.int64), i_id=_t(i_id, torch.int64),
        reuid=_t(reuid, torch.int64), reiid=_t(reiid, torch.int64),
        ratings=_t(ratings),
    )


def datt_batch(cfg, seed=1, edge_cases=False):
    rng = np.random.default_rng(seed)
    c = cfg
    u = _docs(rng, c["B"], c["L"], c["V"])
    i = _docs(rng, c["B"], c["L"], c["V"])
    if edge_cases:
        u[0] = 0
    ratings = rng.integers(1, 6, size=c["B"]).astype(np.float32)
    return dict(u_docs=_t(u, torch.int64), i_docs=_t(i, torch.int64), ratings=_t(ratings))


def siamese_batch(cfg, seed=1, edge_cases=False):
    """u_revs/i_revs [B,R,T] int64, word masks [B,R,T] bool, review masks [B,R] bool, ids [B], ratings."""
    rng = np.random.default_rng(seed)
    c = cfg
    B, R, T, V = c["B"], c["R"], c["T"], c["V"]

    def side():
        txt = _docs(rng, B * R, T, V).reshape(B, R, T)
        nrev = rng.integers(1, R + 1, size=B)
        dead = np.arange(R)[None, :] >= nrev[:, None]     # trailing reviews: all pad
        txt[dead] = 0
        return txt

    u_revs, i_revs = side(), side()
    u_ids = rng.integers(1, c["U"], size=B).astype(np.int64)
    i_ids = rng.integers(1, c["I"], size=B).astype(np.int64)
    ratings = rng.integers(1, 6, size=B).astype(np.float32)
    if edge_cases:
        u_revs[0] = 0               # a user with no review at all: uniform attention over empty bags
        u_ids[0] = 0
    return dict(
        u_revs=_t(u_revs, torch.int64), i_revs=_t(i_revs, torch.int64),
        u_word_masks=_t(u_revs != 0, torch.bool), i_word_masks=_t(i_revs != 0, torch.bool),
        u_rev_masks=_t((u_revs != 0).any(-1), torch.bool), i_rev_masks=_t((i_revs != 0).any(-1), torch.bool),
        u_ids=_t(u_ids, torch.int64), i_ids=_t(i_ids, torch.int64), ratings=_t(ratings),
    )
