"""CPU ORACLE -- test infrastructure, NOT product code.

A CPU restatement (torch fp32 ops on the host) of the reference's review-encoder
hot path, function by function, each citing the reference file:line it follows.
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
this module; the product package (review-based-recommender_amd/) never does and
fails loudly when its HIP library is missing.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against golden vectors produced by running the reference itself
(tests/golden/make_golden.py, build container only) -- forward outputs, the
trainer-step gradients, the clipped norm and parameters after 1 and 3 Adam
steps.  The reference ships no tests or fixtures of its own, so
those generated vectors are the only pin.

All functions take the reference's `state_dict` (a mapping name -> tensor) so the
same dictionary drives the reference modules, this oracle and the HIP modules.
Autograd on these functions is the gradient oracle.
"""
from __future__ import annotations

from typing import Mapping, Sequence

import torch
import torch.nn.functional as F

Params = Mapping[str, torch.Tensor]


# --------------------------------------------------------------------------- layers
def word_embedding(table: torch.Tensor, ids: torch.Tensor, padding_idx: int = 0) -> torch.Tensor:
    """models/deepconn/layers.py:9-24 -- nn.Embedding(V, D, padding_idx=0) row gather.

    padding_idx only affects the gradient (row `padding_idx` receives none)."""
    return F.embedding(ids, table, padding_idx=padding_idx)


def masked_tensor(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """models/deepconn/utils.py:49-61 -- zero rows whose mask is False."""
    assert x.shape[:-1] == mask.shape
    return x.masked_fill(~mask.unsqueeze(-1), 0.0)


def parse_kernel_sizes(kernel_sizes) -> list:
    """models/deepconn/layers.py:34-39 -- list or "3,5,7" string; all odd."""
    if isinstance(kernel_sizes, str):
        kernel_sizes = [int(x) for x in kernel_sizes.split(",")]
    ks = [int(k) for k in kernel_sizes]
    assert all(k % 2 == 1 for k in ks)
    return ks


def my_conv1d(x_ncl: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    """models/deepconn/layers.py:46-60 -- per width 'same' cross-correlation, cat on channels
    (width-major channel order)."""
    outs = []
    for w, b in zip(weights, biases):
        kz = w.shape[-1]
        outs.append(F.conv1d(x_ncl, w, b, padding=(kz - 1) // 2))
    return torch.cat(outs, dim=1)


def conv_params(p: Params, prefix: str = "ngram.feature_layer.0.list_of_conv1d"):
    ws, bs = [], []
    i = 0
    while f"{prefix}.{i}.weight" in p:
        ws.append(p[f"{prefix}.{i}.weight"])
        bs.append(p[f"{prefix}.{i}.bias"])
        i += 1
    return ws, bs


def ngram_feat_cnn(x_nlc: torch.Tensor, mask: torch.Tensor, weights, biases) -> torch.Tensor:
    """models/deepconn/layers.py:100-136, arch="CNN".

    mask -> transpose to NCL -> multi-width conv -> ReLU -> max over ALL seq_len positions
    (pads included: an all-pad document yields relu(bias)).  Returns [N, H]."""
    x = masked_tensor(x_nlc, mask).transpose(1, 2)
    y = F.relu(my_conv1d(x, weights, biases))
    return F.max_pool1d(y, y.shape[-1]).squeeze(-1)


def ngram_feat_hier(x_nlc: torch.Tensor, mask: torch.Tensor, kernel_size: int, proj_w=None, proj_b=None) -> torch.Tensor:
    """models/deepconn/layers.py:62-98,110-114, arch="HierPooling".

    avg_pool1d(k, stride 1) over the masked, transposed input -> global max -> optional
    Linear(D->H) -> ReLU."""
    x = masked_tensor(x_nlc, mask).transpose(1, 2)
    x = F.avg_pool1d(x, kernel_size, stride=1)
    x = F.max_pool1d(x, x.shape[-1]).squeeze(2)
    if proj_w is not None:
        x = F.linear(x, proj_w, proj_b)
    return F.relu(x)


def last_feat(text_feat: torch.Tensor, my_id: torch.Tensor, W, b, ebd) -> torch.Tensor:
    """models/deepconn/layers.py:156-165 -- text_feat @ W + b + ebd[id] (ebd has padding_idx=0)."""
    return text_feat @ W + b + F.embedding(my_id, ebd, padding_idx=0)


def fm(u_feat, i_feat, u_id, i_id, h, g_bias, user_bias, item_bias, dropout_p: float = 0.0, training: bool = False):
    """models/deepconn/layers.py:189-209 -- relu(u*i) -> dropout -> @h + bu[uid] + bi[iid] + g."""
    z = F.relu(u_feat * i_feat)
    z = F.dropout(z, dropout_p, training)
    return z @ h + F.embedding(u_id, user_bias, padding_idx=0) + F.embedding(i_id, item_bias, padding_idx=0) + g_bias


def linear_attention(feat, other_id, W_rv, W_id, h, b_1, b_2, ebd_vals, dropout_p: float = 0.0, training: bool = False):
    """models/narre/narre.py:40-64 -- review-level attention.

    Un-masked, un-stabilised softmax: exp(l) / (sum_R exp(l) + 1e-8).  Returns (out [B,H], att [B,R,1])."""
    e = F.embedding(other_id, ebd_vals, padding_idx=0)
    logits = F.relu(feat @ W_rv + e @ W_id + b_1) @ h + b_2
    ex = logits.exp()
    att = ex / (ex.sum(dim=1, keepdim=True) + 1e-8)
    out = torch.sum(att * feat, dim=1)
    return F.dropout(out, dropout_p, training), att


# --------------------------------------------------------------------------- models
def deepconn_forward(p: Params, u_docs, i_docs, u_masks, i_masks, u_ids, i_ids, arch: str = "CNN",
                     kernel_size: int = 3, dropout_p: float = 0.0, training: bool = False, return_feats: bool = False):
    """models/deepconn/deepconn.py:28-53 -- shared table, SHARED ngram for both towers."""
    table = p["word_embeddings.embedding.weight"]
    ue, ie = word_embedding(table, u_docs), word_embedding(table, i_docs)
    if arch == "CNN":
        ws, bs = conv_params(p)
        uf, itf = ngram_feat_cnn(ue, u_masks, ws, bs), ngram_feat_cnn(ie, i_masks, ws, bs)
    elif arch == "HierPooling":
        pw = p.get("ngram.feature_layer.0.proj_layer.weight")
        pb = p.get("ngram.feature_layer.0.proj_layer.bias")
        uf = ngram_feat_hier(ue, u_masks, kernel_size, pw, pb)
        itf = ngram_feat_hier(ie, i_masks, kernel_size, pw, pb)
    else:
        raise ValueError(f"{arch} is not predefined.")
    ul = last_feat(uf, u_ids, p["user_feat.W"], p["user_feat.b"], p["user_feat.ebd.weight"])
    il = last_feat(itf, i_ids, p["item_feat.W"], p["item_feat.b"], p["item_feat.ebd.weight"])
    pred = fm(ul, il, u_ids, i_ids, p["fm.h"], p["fm.g_bias"], p["fm.user_bias.weight"],
              p["fm.item_bias.weight"], dropout_p, training).view(-1)
    if return_feats:
        return pred, uf, itf
    return pred


def narre_forward(p: Params, u_text, i_text, u_masks, i_masks, u_id, i_id, reuid, reiid,
                  dropout_p: float = 0.0, training: bool = False):
    """models/narre/narre.py:165-192 -- reviews folded into the batch for the shared TextCNN,
    per-side review attention, LastFeat, FM.  Returns (pred [B], u_att [B,R,1], i_att [B,R,1])."""
    table = p["word_embeddings.embedding.weight"]
    B, R, T = u_text.shape
    ws, bs = conv_params(p)

    def tower(text, masks, other_id, att):
        e = word_embedding(table, text).view(B * R, T, -1)
        f = ngram_feat_cnn(e, masks.view(B * R, T), ws, bs).view(B, R, -1)
        return linear_attention(f, other_id, p[f"{att}.W_rv"], p[f"{att}.W_id"], p[f"{att}.h"], p[f"{att}.b_1"],
                                p[f"{att}.b_2"], p[f"{att}.ebd_vals.weight"], dropout_p, training)

    uf, ua = tower(u_text, u_masks, reuid, "user_att")
    itf, ia = tower(i_text, i_masks, reiid, "item_att")
    ul = last_feat(uf, u_id, p["user_feat.W"], p["user_feat.b"], p["user_feat.ebd.weight"])
    il = last_feat(itf, i_id, p["item_feat.W"], p["item_feat.b"], p["item_feat.ebd.weight"])
    pred = fm(ul, il, u_id, i_id, p["fm.h"], p["fm.g_bias"], p["fm.user_bias.weight"],
              p["fm.item_bias.weight"], dropout_p, training)
    return pred.view(-1), ua, ia


def local_attention(x_nlc, attn_w, attn_b, conv_w, conv_b):
    """models/dual_att/layers.py:43-53 -- sigmoid(conv1d(E,1,win,'same')) gate per token,
    x*gate -> conv1d(E,out,1) -> tanh -> max over all L.  No masks.  Returns [B,out,1]."""
    x = x_nlc.permute(0, 2, 1)
    win = attn_w.shape[-1]
    score = torch.sigmoid(F.conv1d(x, attn_w, attn_b, padding=(win - 1) // 2))
    y = torch.tanh(F.conv1d(score * x, conv_w, conv_b))
    return F.max_pool1d(y, y.shape[-1])


def global_attention(x_nlc, attn_w, attn_b, convs):
    """models/dual_att/layers.py:81-89 -- ONE sigmoid scalar per doc (kernel = doc_len), x*gate,
    then 'valid' convs k=2,3,4 -> tanh -> max over L-k+1.  Returns 3 x [B,out,1]."""
    x = x_nlc.permute(0, 2, 1)
    score = torch.sigmoid(F.conv1d(x, attn_w, attn_b))     # [B,1,1]
    g = score * x
    outs = []
    for w, b in convs:
        y = torch.tanh(F.conv1d(g, w, b))
        outs.append(F.max_pool1d(y, y.shape[-1]))
    return tuple(outs)


def datt_forward(p: Params, u_docs, i_docs, dropout_p: float = 0.0, training: bool = False):
    """models/dual_att/dual_att.py:37-61 -- per-tower local+global attention, ONE shared fc
    (Linear, ReLU, Dropout, Linear), ratings = sum(u*i)."""
    table = p["word_embeddings.embedding.weight"]

    def tower(docs, s):
        x = word_embedding(table, docs)
        lo = local_attention(x, p[f"{s}_local_atten.attn.0.weight"], p[f"{s}_local_atten.attn.0.bias"],
                             p[f"{s}_local_atten.conv.0.weight"], p[f"{s}_local_atten.conv.0.bias"])
        g = f"{s}_global_atten"
        go = global_attention(x, p[f"{g}.attn.0.weight"], p[f"{g}.attn.0.bias"],
                              [(p[f"{g}.conv{n}.0.weight"], p[f"{g}.conv{n}.0.bias"]) for n in (1, 2, 3)])
        feat = torch.cat((lo,) + go, 1).flatten(1)
        hdn = F.dropout(F.relu(F.linear(feat, p["fc.0.weight"], p["fc.0.bias"])), dropout_p, training)
        return F.linear(hdn, p["fc.3.weight"], p["fc.3.bias"])

    return torch.sum(tower(u_docs, "u") * tower(i_docs, "i"), 1).view(-1)


# --------------------------------------------------------------------------- train step
# --------------------------------------------------------------------------- SimpleSiamese
def masked_avg_pool(x_nlc: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """models/simple_siamese/layers.py:90-110 -- sum of the unmasked rows / (count + 1e-8).  [n, D]."""
    fm_ = mask.unsqueeze(-1).float()
    return (x_nlc * fm_).sum(dim=1) / (fm_.sum(dim=1) + 1e-8)


def additive_attention(x, mask, proj_w, proj_b, inner_w):
    """models/simple_siamese/layers.py:171-197 -- logits = inner(tanh(proj(x))); masked softmax over the reviews
    (masked_fill -1e8: a row with no valid review becomes uniform); weighted sum.  Returns ([B,H], [B,R,1])."""
    logits = F.linear(torch.tanh(F.linear(x, proj_w, proj_b)), inner_w)            # [B,R,1]
    scores = F.softmax(torch.masked_fill(logits, ~mask.unsqueeze(2), -1e8), dim=1)
    return torch.sum(scores * x, dim=1), scores


def siamese_forward(p: Params, u_revs, i_revs, u_word_masks, i_word_masks, u_rev_masks, i_rev_masks, u_ids, i_ids,
                    dropout_p: float = 0.0, word_dropout_p: float = 0.0, review_dropout_p: float = 0.0,
                    training: bool = False):
    """models/simple_siamese/simple_siamese.py:38-87 -- bag-of-embeddings reviews (variational word dropout, masked
    average), optional Linear+Tanh, review (node) dropout, additive attention over reviews shared by both towers,
    LastFeat, FM (with or without user / item biases).  Returns (pred [B], u_scores, i_scores)."""
    table = p["word_embedding.embedding.weight"]
    B = u_revs.shape[0]
    lt = "latent_transform_layer.0.weight" in p

    def tower(revs, wmask, rmask):
        _, R, T = revs.shape
        e = word_embedding(table, revs).view(B * R, T, -1)
        if training and word_dropout_p > 0:                                     # layers.py:24-50: one mask per (review, dim)
            e = e * F.dropout(torch.ones(B * R, e.shape[-1]), word_dropout_p, True).unsqueeze(1)
        r = masked_avg_pool(e, wmask.view(B * R, T)).view(B, R, -1)
        if lt:
            r = torch.tanh(F.linear(r, p["latent_transform_layer.0.weight"], p["latent_transform_layer.0.bias"]))
        if training and review_dropout_p > 0:                                   # layers.py:7-22: one mask per review
            r = r * F.dropout(torch.ones(B, R), review_dropout_p, True).unsqueeze(2)
        return additive_attention(r, rmask, p["review_att_layer.proj_layer.0.weight"],
                                  p["review_att_layer.proj_layer.0.bias"], p["review_att_layer.inner_product.weight"])

    uf, us = tower(u_revs, u_word_masks, u_rev_masks)
    itf, is_ = tower(i_revs, i_word_masks, i_rev_masks)
    ul = last_feat(uf, u_ids, p["user_last_feat_layer.W"], p["user_last_feat_layer.b"], p["user_last_feat_layer.ebd.weight"])
    il = last_feat(itf, i_ids, p["item_last_feat_layer.W"], p["item_last_feat_layer.b"], p["item_last_feat_layer.ebd.weight"])
    if "fm.user_bias.weight" in p:
        pred = fm(ul, il, u_ids, i_ids, p["fm.h"], p["fm.g_bias"], p["fm.user_bias.weight"], p["fm.item_bias.weight"],
                  dropout_p, training)
    else:                                                                       # FMWithoutUIBias, layers.py:263-297
        z = F.dropout(F.relu(ul * il), dropout_p, training)
        pred = z @ p["fm.h"] + p["fm.g_bias"]
    return pred.view(-1), us, is_


def train_steps(params: Params, forward_fn, ratings, n_steps: int = 1, lr: float = 2e-3, max_grad_norm: float = 5.0):
    """trainer/train_deepconn_pp.py:161-168 -- zero_grad, forward, MSELoss(mean), backward,
    clip_grad_norm_(max_grad_norm), Adam(lr).  `forward_fn(p) -> pred`.

    Returns a list (one dict per step) of pred, loss, grads (pre-clip), gnorm, and the leaf
    parameters (updated in place)."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=lr)
    history = []
    for _ in range(n_steps):
        opt.zero_grad()
        pred = forward_fn(leaves)
        loss = F.mse_loss(pred, ratings)
        loss.backward()
        grads = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
        gnorm = torch.nn.utils.clip_grad_norm_(list(leaves.values()), max_grad_norm)
        opt.step()
        history.append(dict(pred=pred.detach().clone(), loss=loss.detach().clone(), grads=grads,
                            gnorm=gnorm.detach().clone(),
                            params={k: v.detach().clone() for k, v in leaves.items()}))
    return history


def train_over_batches(params: Params, steps, lr: float = 2e-3, max_grad_norm: float = 5.0):
    """The same step as train_steps() over a SEQUENCE of batches (one epoch of trainer/train_deepconn_pp.py:143-168):
    `steps` = [(forward_fn(p) -> pred, ratings), ...].  Returns (per-step losses, the trained leaf parameters)."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=lr)
    losses = []
    for forward_fn, ratings in steps:
        opt.zero_grad()
        loss = F.mse_loss(forward_fn(leaves), ratings)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(leaves.values()), max_grad_norm)
        opt.step()
        losses.append(float(loss))
    return losses, {k: v.detach() for k, v in leaves.items()}
