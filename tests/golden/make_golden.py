#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by running the REFERENCE.

Runs only in the build container (needs /root/reference, CPU only):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's nn.Modules (models/deepconn/deepconn.py:10, models/narre/narre.py:139,
models/dual_att/layers.py:25,55) are imported read-only, loaded with the seeded
state_dicts from synth.py, and driven through the trainer's step
(trainer/train_deepconn_pp.py:161-168: zero_grad -> forward -> MSELoss -> backward ->
clip_grad_norm_(5.0) -> Adam(lr=2e-3)).  Only inputs-by-seed and OUTPUT tensors are
stored: no reference source or bytecode is written anywhere.

models/dual_att/dual_att.py cannot be imported here (its unused top-level
`from nltk import word_tokenize` raises ModuleNotFoundError; nltk is absent and is not
stubbed).  The D-ATT vectors therefore come from the reference's own
LocalAttention / GlobalAttention / WordEmbedding classes (models/dual_att/layers.py,
which import fine) wired together by `_RefDualAttWiring` below exactly as
dual_att.py:26-61 describes (one shared fc, dot product of tower outputs).
"""
from __future__ import annotations

import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import synth  # noqa: E402

LR = 2e-3          # default_deepconn_pp.json:24
MAX_GNORM = 5.0    # default_deepconn_pp.json:27
BIG = {"cfg1", "cfg2", "cfg3", "cfg4"}


def _quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _np(t):
    return t.detach().cpu().numpy().copy()   # copy: clip_grad_norm_ later scales .grad in place


def _sample(flat: np.ndarray, n=4096):
    """Deterministic strided sample of a big gradient (kept small in the fixture)."""
    flat = flat.reshape(-1)
    if flat.size <= n:
        return flat.copy()
    step = flat.size // n
    return flat[::step][:n].copy()


def run_train_steps(model, fwd, ratings, out, big, n_steps=3):
    """The a-13 step: records pred/loss/grads/gnorm + params after steps 1 and 3."""
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    loss_fn = nn.MSELoss()
    model.train()
    for step in range(n_steps):
        opt.zero_grad()
        pred = fwd()
        loss = loss_fn(pred, ratings)
        loss.backward()
        if step == 0:
            out["pred"] = _np(pred)
            out["loss"] = _np(loss)
            for k, p in model.named_parameters():
                g = _np(p.grad)
                out[f"gradl2/{k}"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                out[f"gradabs/{k}"] = np.float64(np.abs(g.astype(np.float64)).sum())
                if big and g.size > 65536:
                    out[f"gradsample/{k}"] = _sample(g)
                else:
                    out[f"grad/{k}"] = g
        gnorm = nn.utils.clip_grad_norm_(model.parameters(), MAX_GNORM)
        if step == 0:
            out["gnorm"] = _np(gnorm)
        opt.step()
        if step in (0, 2):
            tag = f"after{step + 1}"
            for k, p in model.named_parameters():
                v = _np(p)
                if big and v.size > 65536:
                    out[f"{tag}sample/{k}"] = _sample(v)
                else:
                    out[f"{tag}/{k}"] = v
    out["loss_after3"] = _np(loss)


# ----------------------------------------------------------------------------- DeepCoNN
def gen_deepconn(name, arch="CNN", edge=False, seed=0):
    from models.deepconn.deepconn import DeepCoNNpp
    cfg = synth.DEEPCONN_CFGS[name]
    c = cfg
    big = name in BIG
    sd = synth.deepconn_params(cfg, seed) if arch == "CNN" else synth.deepconn_hier_params(cfg, seed)
    kz = c["kz"] if arch == "CNN" else c["kz"][:1]
    model = _quiet(DeepCoNNpp, c["U"], c["I"], c["V"], kz, c["D"], c["H"], c["K"], c["L"], None, 0.0, arch)
    model.load_state_dict(sd)
    b = synth.deepconn_batch(cfg, seed + 1, edge_cases=edge)
    out = {}
    feats = {}
    def _grab(_m, _i, o):          # returns None so the module output is left untouched
        feats[len(feats)] = _np(o)

    hook = model.ngram.register_forward_hook(_grab)
    model.eval()
    with torch.no_grad():
        out["pred_eval"] = _np(model(b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"]))
    hook.remove()
    out["u_rev_feats"] = feats[0].reshape(c["B"], c["H"])
    out["i_rev_feats"] = feats[1].reshape(c["B"], c["H"])
    run_train_steps(
        model,
        lambda: model(b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"]),
        b["ratings"], out, big)
    return out


# ----------------------------------------------------------------------------- NARRE
def gen_narre(name, edge=False, seed=0):
    from models.narre.narre import NARRE
    cfg = synth.NARRE_CFGS[name]
    c = cfg
    big = name in BIG
    model = _quiet(NARRE, c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"],
                   c["R"], c["T"], 0.0, 0, 0, 0, None, "CNN")
    model.load_state_dict(synth.narre_params(cfg, seed))
    b = synth.narre_batch(cfg, seed + 1, edge_cases=edge)
    args = (b["u_text"], b["i_text"], b["u_masks"], b["i_masks"], b["u_id"], b["i_id"], b["reuid"], b["reiid"])
    out = {}
    model.eval()
    with torch.no_grad():
        pred, ua, ia = model(*args)
    out["pred_eval"], out["u_att"], out["i_att"] = _np(pred), _np(ua), _np(ia)
    run_train_steps(model, lambda: model(*args)[0], b["ratings"], out, big)
    return out


# ----------------------------------------------------------------------------- D-ATT
class _RefDualAttWiring(nn.Module):
    """The reference's own layer classes, wired as dual_att.py:26-61 (see module docstring)."""

    def __init__(self, c):
        super().__init__()
        from models.dual_att.layers import GlobalAttention, LocalAttention, WordEmbedding
        self.word_embeddings = WordEmbedding(c["V"], c["E"], pretrained_embeddings=None)
        self.u_local_atten = LocalAttention(c["L"], c["win"], c["l_out"], c["E"])
        self.u_global_atten = GlobalAttention(c["L"], c["g_out"], c["E"])
        self.i_local_atten = LocalAttention(c["L"], c["win"], c["l_out"], c["E"])
        self.i_global_atten = GlobalAttention(c["L"], c["g_out"], c["E"])
        self.fc = nn.Sequential(nn.Linear(c["l_out"] + 3 * c["g_out"], c["h1"]), nn.ReLU(),
                                nn.Dropout(0.0), nn.Linear(c["h1"], c["h2"]))

    def tower(self, x, loc, glo):
        x = self.word_embeddings(x)
        feat = torch.cat((loc(x),) + tuple(glo(x)), 1)
        return self.fc(feat.view(feat.size(0), -1))

    def forward(self, u_docs, i_docs):
        u = self.tower(u_docs, self.u_local_atten, self.u_global_atten)
        i = self.tower(i_docs, self.i_local_atten, self.i_global_atten)
        return torch.sum(u * i, 1).view(-1)


def gen_datt(name, edge=False, seed=0):
    cfg = synth.DATT_CFGS[name]
    big = name in BIG
    model = _quiet(_RefDualAttWiring, cfg)
    scale = 0.3 if big else 1.0
    model.load_state_dict(synth.datt_params(cfg, seed, table_scale=scale))
    b = synth.datt_batch(cfg, seed + 1, edge_cases=edge)
    out = {}
    model.eval()
    with torch.no_grad():
        out["pred_eval"] = _np(model(b["u_docs"], b["i_docs"]))
    run_train_steps(model, lambda: model(b["u_docs"], b["i_docs"]), b["ratings"], out, big)
    return out


# ----------------------------------------------------------------------------- SimpleSiamese
def gen_siamese(name, edge=False, seed=0):
    from models.simple_siamese.simple_siamese import SimpleSiamese
    c = synth.SIAMESE_CFGS[name]
    # dropout rates 0: the fixtures pin the deterministic math (dropout draws cannot match across generators)
    model = _quiet(SimpleSiamese, c["D"], c["K"], c["V"], c["U"], c["I"], None, False, 0.0, 0.0, 0.0, c["UB"], c["LT"])
    model.load_state_dict(synth.siamese_params(c, seed))
    b = synth.siamese_batch(c, seed + 1, edge_cases=edge)
    args = tuple(b[k] for k in ("u_revs", "i_revs", "u_word_masks", "i_word_masks", "u_rev_masks", "i_rev_masks", "u_ids", "i_ids"))
    out = {}
    scores = []
    hook = model.review_att_layer.register_forward_hook(lambda _m, _i, o: scores.append(_np(o[1])) and None)
    model.eval()
    with torch.no_grad():
        out["pred_eval"] = _np(model(*args)[0])
    hook.remove()
    out["u_rev_scores"] = scores[0].reshape(c["B"], c["R"])
    out["i_rev_scores"] = scores[1].reshape(c["B"], c["R"])
    run_train_steps(model, lambda: model(*args)[0], b["ratings"], out, big=(name == "toys"))
    return out


def main():
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    jobs = [
        ("deepconn_tiny", lambda: gen_deepconn("tiny", edge=True)),
        ("deepconn_small", lambda: gen_deepconn("small", edge=True)),
        ("deepconn_k3", lambda: gen_deepconn("k3", edge=False)),
        ("deepconn_hier_small", lambda: gen_deepconn("small", arch="HierPooling", edge=True)),
        ("deepconn_cfg1", lambda: gen_deepconn("cfg1")),
        ("deepconn_cfg2", lambda: gen_deepconn("cfg2")),
        ("narre_tiny", lambda: gen_narre("tiny", edge=True)),
        ("narre_small", lambda: gen_narre("small", edge=True)),
        ("narre_cfg3", lambda: gen_narre("cfg3")),
        ("datt_tiny", lambda: gen_datt("tiny", edge=True)),
        ("datt_small", lambda: gen_datt("small", edge=True)),
        ("datt_cfg4", lambda: gen_datt("cfg4")),
        ("siamese_tiny", lambda: gen_siamese("tiny", edge=True)),
        ("siamese_small", lambda: gen_siamese("small", edge=True)),
        ("siamese_toys", lambda: gen_siamese("toys")),
    ]
    for name, fn in jobs:
        if only and name not in only:
            continue
        out = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
