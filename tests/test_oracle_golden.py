"""Pins oracle/ref_cpu.py against the golden vectors captured from the reference
(tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import ref_cpu as O

FWD_TOL = 2e-6     # same ATen kernels on the same host: only reduction-order noise is expected
GRAD_RTOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    assert (err <= lim).all(), f"{what}: max err {err.max():.3e} (limit {lim.min():.3e})"


def _check_steps(g, hist, big):
    h0 = hist[0]
    _close(h0["pred"], g["pred"], FWD_TOL, what="pred(train)")
    _close(h0["loss"], g["loss"], 1e-5, what="loss")
    _close(h0["gnorm"], g["gnorm"], 0, 1e-5, what="gnorm")
    for k, gr in h0["grads"].items():
        gn = gr.numpy()
        scale = float(g[f"gradl2/{k}"]) + 1e-12
        l2 = float(np.sqrt((gn.astype(np.float64) ** 2).sum()))
        assert abs(l2 - scale) <= GRAD_RTOL * scale + 1e-9, (k, l2, scale)
        if f"grad/{k}" in g:
            _close(gn, g[f"grad/{k}"], 1e-6 + GRAD_RTOL * scale, what=f"grad {k}")
        else:
            from make_golden import _sample
            _close(_sample(gn), g[f"gradsample/{k}"], 1e-6 + GRAD_RTOL * scale, what=f"gradsample {k}")
    for step, tag in ((0, "after1"), (2, "after3")):
        if len(hist) <= step:
            continue
        for k, v in hist[step]["params"].items():
            if float(g[f"gradl2/{k}"]) < 1e-6:
                # analytically-zero gradient (NARRE's b_2: softmax is shift invariant up to the
                # 1e-8 term) -> Adam normalises rounding noise into +-lr steps; nothing to pin
                continue
            if f"{tag}/{k}" in g:
                # Adam's first steps move every touched weight by ~lr whatever the gradient size:
                # a sign flip of a ~0 gradient is the only way to differ by more than this
                _close(v.numpy(), g[f"{tag}/{k}"], 5e-5, what=f"{tag} {k}")


@pytest.mark.parametrize("name,cfgname,edge", [
    ("deepconn_tiny", "tiny", True), ("deepconn_small", "small", True), ("deepconn_k3", "k3", False),
    ("deepconn_cfg1", "cfg1", False)])
def test_deepconn_oracle_matches_reference(golden_dir, name, cfgname, edge):
    g = _load(golden_dir, name)
    cfg = synth.DEEPCONN_CFGS[cfgname]
    p = synth.deepconn_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1, edge_cases=edge)
    args = (b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"])
    with torch.no_grad():
        pred, uf, itf = O.deepconn_forward(p, *args, return_feats=True)
    _close(pred, g["pred_eval"], FWD_TOL, what="pred_eval")
    _close(uf, g["u_rev_feats"], FWD_TOL, what="u_rev_feats")
    _close(itf, g["i_rev_feats"], FWD_TOL, what="i_rev_feats")
    hist = O.train_steps(p, lambda q: O.deepconn_forward(q, *args), b["ratings"], n_steps=3)
    _check_steps(g, hist, cfgname.startswith("cfg"))


def test_deepconn_hierpooling_oracle(golden_dir):
    g = _load(golden_dir, "deepconn_hier_small")
    cfg = synth.DEEPCONN_CFGS["small"]
    p = synth.deepconn_hier_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1, edge_cases=True)
    args = (b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"])
    kw = dict(arch="HierPooling", kernel_size=cfg["kz"][0])
    with torch.no_grad():
        _close(O.deepconn_forward(p, *args, **kw), g["pred_eval"], FWD_TOL, what="pred_eval")
    hist = O.train_steps(p, lambda q: O.deepconn_forward(q, *args, **kw), b["ratings"], n_steps=3)
    _check_steps(g, hist, False)


def test_deepconn_cfg2_forward_oracle(golden_dir):
    """Full BASELINE size (B=256, 2x512 tokens, D=300, widths 3/5/7): forward + loss only (~1 s)."""
    g = _load(golden_dir, "deepconn_cfg2")
    cfg = synth.DEEPCONN_CFGS["cfg2"]
    p = synth.deepconn_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1)
    with torch.no_grad():
        pred = O.deepconn_forward(p, b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"])
    _close(pred, g["pred_eval"], 1e-5, what="pred_eval")
    mse = torch.mean((pred - b["ratings"]) ** 2)
    _close(mse, g["loss"], 1e-4, what="MSE")


@pytest.mark.parametrize("name,cfgname,edge", [("narre_tiny", "tiny", True), ("narre_small", "small", True)])
def test_narre_oracle_matches_reference(golden_dir, name, cfgname, edge):
    g = _load(golden_dir, name)
    cfg = synth.NARRE_CFGS[cfgname]
    p = synth.narre_params(cfg, 0)
    b = synth.narre_batch(cfg, 1, edge_cases=edge)
    args = (b["u_text"], b["i_text"], b["u_masks"], b["i_masks"], b["u_id"], b["i_id"], b["reuid"], b["reiid"])
    with torch.no_grad():
        pred, ua, ia = O.narre_forward(p, *args)
    _close(pred, g["pred_eval"], FWD_TOL, what="pred_eval")
    _close(ua, g["u_att"], FWD_TOL, what="u_att")
    _close(ia, g["i_att"], FWD_TOL, what="i_att")
    hist = O.train_steps(p, lambda q: O.narre_forward(q, *args)[0], b["ratings"], n_steps=3)
    _check_steps(g, hist, False)


def test_narre_cfg3_forward_oracle(golden_dir):
    g = _load(golden_dir, "narre_cfg3")
    cfg = synth.NARRE_CFGS["cfg3"]
    p = synth.narre_params(cfg, 0)
    b = synth.narre_batch(cfg, 1)
    with torch.no_grad():
        pred, ua, ia = O.narre_forward(p, b["u_text"], b["i_text"], b["u_masks"], b["i_masks"], b["u_id"], b["i_id"],
                                       b["reuid"], b["reiid"])
    _close(pred, g["pred_eval"], 1e-5, what="pred_eval")
    _close(ua, g["u_att"], 1e-5, what="u_att")


@pytest.mark.parametrize("name,cfgname", [("datt_tiny", "tiny"), ("datt_small", "small")])
def test_datt_oracle_matches_reference(golden_dir, name, cfgname):
    g = _load(golden_dir, name)
    cfg = synth.DATT_CFGS[cfgname]
    p = synth.datt_params(cfg, 0)
    b = synth.datt_batch(cfg, 1, edge_cases=True)
    with torch.no_grad():
        _close(O.datt_forward(p, b["u_docs"], b["i_docs"]), g["pred_eval"], FWD_TOL, what="pred_eval")
    hist = O.train_steps(p, lambda q: O.datt_forward(q, b["u_docs"], b["i_docs"]), b["ratings"], n_steps=3)
    _check_steps(g, hist, False)


SIAMESE_ARGS = ("u_revs", "i_revs", "u_word_masks", "i_word_masks", "u_rev_masks", "i_rev_masks", "u_ids", "i_ids")


@pytest.mark.parametrize("name,cfgname,edge", [("siamese_tiny", "tiny", True), ("siamese_small", "small", True),
                                               ("siamese_toys", "toys", False)])
def test_siamese_oracle_matches_reference(golden_dir, name, cfgname, edge):
    """SimpleSiamese: with / without latent transform and user / item biases."""
    g = _load(golden_dir, name)
    cfg = synth.SIAMESE_CFGS[cfgname]
    p = synth.siamese_params(cfg, 0)
    b = synth.siamese_batch(cfg, 1, edge_cases=edge)
    args = tuple(b[k] for k in SIAMESE_ARGS)
    with torch.no_grad():
        pred, us, is_ = O.siamese_forward(p, *args)
    _close(pred, g["pred_eval"], FWD_TOL, what="pred_eval")
    _close(us.view(cfg["B"], cfg["R"]), g["u_rev_scores"], FWD_TOL, what="u_rev_scores")
    _close(is_.view(cfg["B"], cfg["R"]), g["i_rev_scores"], FWD_TOL, what="i_rev_scores")
    hist = O.train_steps(p, lambda q: O.siamese_forward(q, *args)[0], b["ratings"], n_steps=3)
    _check_steps(g, hist, cfgname == "toys")
