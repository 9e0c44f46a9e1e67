"""Edge cases and full-size properties of the fused TextCNN kernels (through the C ABI), beyond the
golden-vector shapes: scalar-gather path (D % 4 != 0), widths 1 and 9, one document, L < window,
L = 33 (one token into the second slab), > 7 channel tiles (two launches), all-masked batch,
and size-independent properties at the BASELINE cfg2 size."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from helpers import max_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_textcnn(table, ids, mask, ws, bs, gate=None, valid=False, tanh=False):
    x = F.embedding(ids, table)
    if mask is not None:
        x = x.masked_fill(~mask.unsqueeze(-1), 0.0)
    if gate is not None:
        x = x * gate.unsqueeze(-1)
    x = x.transpose(1, 2)
    outs = []
    for w, b in zip(ws, bs):
        k = w.shape[2]
        y = F.conv1d(x, w, b, padding=0 if valid else (k - 1) // 2)
        y = torch.tanh(y) if tanh else F.relu(y)
        outs.append(F.max_pool1d(y, y.shape[-1]).squeeze(-1))
    return torch.cat(outs, 1)


def _case(n_docs, L, D, V, kzs, chans, seed, mask_p=0.2, gate=False, valid=False, tanh=False):
    from review_based_recommender_amd import functional as RF
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(V, D, generator=g)
    ids = torch.randint(0, V, (n_docs, L), generator=g)
    mask = (torch.rand(n_docs, L, generator=g) > mask_p) if mask_p is not None else None
    ws = [torch.randn(c, D, k, generator=g) * (1.0 / np.sqrt(D * k)) for k, c in zip(kzs, chans)]
    bs = [torch.randn(c, generator=g) * 0.1 for c in chans]
    gt = torch.rand(n_docs, L, generator=g) * 0.8 + 0.1 if gate else None
    leaves = [table.clone().requires_grad_(True)] + [w.clone().requires_grad_(True) for w in ws] + \
             [b.clone().requires_grad_(True) for b in bs] + ([gt.clone().requires_grad_(True)] if gate else [])
    nw = len(ws)
    ref = _oracle_textcnn(leaves[0], ids, mask, leaves[1:1 + nw], leaves[1 + nw:1 + 2 * nw],
                          leaves[-1] if gate else None, valid, tanh)
    d_out = torch.randn(ref.shape, generator=g)
    ref.backward(d_out)

    dl = [t.detach().clone().to(DEV).requires_grad_(True) for t in leaves]
    out = RF.textcnn(dl[0], ids.to(DEV), mask.to(DEV) if mask is not None else None, dl[1:1 + nw], dl[1 + nw:1 + 2 * nw],
                     gate=dl[-1] if gate else None, pad_mode=RF.PAD_VALID if valid else RF.PAD_SAME,
                     act=RF.ACT_TANH if tanh else RF.ACT_RELU, padding_idx=None)
    out.backward(d_out.to(DEV))
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 3e-5
    for a, b in zip(dl, leaves):
        scale = float(b.grad.norm()) + 1e-6
        assert max_err(a.grad.cpu().numpy(), b.grad.numpy()) <= 2e-6 + 2e-4 * scale


@pytest.mark.parametrize("shape", [
    dict(n_docs=3, L=20, D=10, V=30, kzs=[3, 5], chans=[4, 4]),            # D % 4 != 0: scalar gather path
    dict(n_docs=2, L=40, D=12, V=25, kzs=[1, 9], chans=[3, 5]),            # narrowest and widest windows
    dict(n_docs=1, L=1, D=8, V=5, kzs=[3], chans=[2]),                      # one token, window wider than the doc
    dict(n_docs=5, L=33, D=16, V=40, kzs=[3, 7], chans=[33, 31]),           # second slab holds a single token
    dict(n_docs=4, L=64, D=20, V=50, kzs=[3], chans=[260]),                 # 9 channel tiles -> two launches
    dict(n_docs=3, L=48, D=24, V=50, kzs=[3, 5, 7], chans=[10, 10, 10], mask_p=1.1),   # every token masked
    dict(n_docs=3, L=48, D=24, V=50, kzs=[3, 5], chans=[6, 6], mask_p=None),            # mask == NULL
    dict(n_docs=3, L=30, D=104, V=60, kzs=[2, 3, 4], chans=[20, 20, 20], mask_p=None, gate=True, valid=True, tanh=True),
    dict(n_docs=2, L=37, D=60, V=60, kzs=[1], chans=[70], mask_p=None, gate=True, tanh=True),   # D-ATT local conv shape
    # short contractions with >= 4 column groups: the rows-stationary GEMM (prod_gemm_b16k_kernel), 4 .. 8 K steps, row blocks
    # that are whole (counted waits across the group boundary) and cut by the end of the token list
    dict(n_docs=6, L=200, D=100, V=300, kzs=[2, 3, 4], chans=[100, 100, 100], mask_p=None, gate=True, valid=True, tanh=True),
    dict(n_docs=8, L=160, D=64, V=200, kzs=[3, 5], chans=[64, 64]),
    dict(n_docs=8, L=160, D=72, V=131, kzs=[3, 5], chans=[64, 64]),
    dict(n_docs=8, L=160, D=88, V=257, kzs=[1, 3, 5], chans=[70, 64, 64], mask_p=None),
    dict(n_docs=8, L=256, D=128, V=600, kzs=[3, 5, 7], chans=[40, 40, 40]),
])
def test_textcnn_edge_shapes(shape, conv_mode):
    _case(seed=11, **shape)


def test_unsupported_shapes_raise():
    from review_based_recommender_amd import functional as RF
    t = torch.randn(10, 8, device=DEV)
    ids = torch.zeros(2, 16, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError, match="odd"):
        RF.textcnn(t, ids, None, [torch.randn(4, 8, 4, device=DEV)], [torch.zeros(4, device=DEV)])
    with pytest.raises(RuntimeError, match="unsupported"):
        RF.textcnn(t, ids, None, [torch.randn(4, 8, 11, device=DEV)], [torch.zeros(4, device=DEV)])


def _cfg2_inputs():
    cfg = synth.DEEPCONN_CFGS["cfg2"]
    p = synth.deepconn_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1)
    ws = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.weight"].to(DEV) for i in range(3)]
    bs = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.bias"].to(DEV) for i in range(3)]
    table = p["word_embeddings.embedding.weight"].to(DEV)
    ids = torch.cat([b["u_docs"], b["i_docs"]]).to(DEV)
    mask = torch.cat([b["u_masks"], b["i_masks"]]).to(DEV)
    return table, ids, mask, ws, bs


def test_fullsize_properties_cfg2(conv_mode):
    """At B=256 x 2 x 512 tokens the oracle is too slow for gradients; check properties instead."""
    from review_based_recommender_amd import functional as RF
    table, ids, mask, ws, bs = _cfg2_inputs()
    f1, a1 = RF.textcnn(table, ids, mask, ws, bs, return_argmax=True)
    f2, a2 = RF.textcnn(table, ids, mask, ws, bs, return_argmax=True)
    assert torch.equal(f1, f2) and torch.equal(a1, a2)                       # forward is bitwise reproducible
    perm = torch.randperm(ids.shape[0], device=DEV)
    fp, ap = RF.textcnn(table, ids[perm], mask[perm], ws, bs, return_argmax=True)
    assert torch.equal(fp, f1[perm]) and torch.equal(ap, a1[perm])           # documents are independent
    junk = torch.where(mask, ids, torch.randint_like(ids, 2, 50002))         # masked tokens never matter
    fj = RF.textcnn(table, junk, mask, ws, bs)
    assert torch.equal(fj, f1)
    assert int(a1.min()) >= 0 and int(a1.max()) < 512
    assert float(f1.min()) >= 0.0                                            # ReLU
    # an all-pad document yields relu(bias) (reference quirk 2)
    z_ids, z_mask = torch.zeros_like(ids[:2]), torch.zeros_like(mask[:2])
    fz = RF.textcnn(table, z_ids, z_mask, ws, bs)
    assert torch.allclose(fz[0], torch.relu(torch.cat(bs)), atol=0, rtol=0)


def test_fullsize_backward_properties_cfg2(conv_mode):
    from review_based_recommender_amd import functional as RF
    table, ids, mask, ws, bs = _cfg2_inputs()

    def grads(scale):
        t = table.clone().requires_grad_(True)
        w = [x.clone().requires_grad_(True) for x in ws]
        b = [x.clone().requires_grad_(True) for x in bs]
        f = RF.textcnn(t, ids, mask, w, b)
        g = torch.Generator(device=DEV).manual_seed(5)
        d = torch.randn(f.shape, generator=g, device=DEV) * scale
        f.backward(d)
        return f.detach(), d, t.grad, [x.grad for x in w], [x.grad for x in b]

    f, d, gt, gw, gb = grads(1.0)
    _, _, gt2, gw2, gb2 = grads(2.0)
    # linearity in the upstream gradient (dW / dbias are reduced in a fixed order -> exactly 2x)
    for a, b2 in zip(gw + gb, gw2 + gb2):
        assert torch.equal(a * 2.0, b2)
    assert float((gt * 2.0 - gt2).abs().max()) <= 1e-4 * float(gt2.abs().max())   # atomics: order noise only
    # checksum: dbias[c] = sum over documents of the gradient that passes the ReLU
    act = (f > 0).float() * d
    assert torch.allclose(torch.cat(gb), act.sum(0), rtol=1e-4, atol=1e-4)
    assert float(gt[0].abs().max()) == 0.0                                   # padding_idx row
    # only tokens that occur un-masked in the batch can receive gradient
    seen = torch.zeros(table.shape[0], dtype=torch.bool, device=DEV)
    seen[ids[mask]] = True
    assert float(gt[~seen].abs().max()) == 0.0


def _random_shape(rng):
    """A random TextCNN problem with heavily repeated tokens (small vocabulary): exercises the token-product tables,
    the several dW variants (float4 slots x documents, document-centric, scalar) and multi-group channel layouts."""
    valid = bool(rng.integers(0, 2))
    n_w = int(rng.integers(1, 4))
    kzs = sorted(set(int(k) for k in rng.choice([2, 3, 4] if valid else [1, 3, 5, 7, 9], size=n_w, replace=False)))
    if not valid:
        kzs = [k for k in kzs if k % 2 == 1]
    D = int(rng.choice([4, 8, 12, 36, 60, 100, 152, 300]))
    L = int(rng.integers(max(kzs), 97))
    n_docs = int(rng.choice([1, 2, 5, 17, 70]))
    V = int(rng.choice([7, 50, 400]))
    chans = [int(rng.choice([1, 5, 32, 50, 90])) for _ in kzs]
    return dict(n_docs=n_docs, L=L, D=D, V=V, kzs=kzs, chans=chans, mask_p=[None, 0.0, 0.3, 0.8][int(rng.integers(0, 4))],
                gate=bool(rng.integers(0, 2)), valid=valid, tanh=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("case", range(24))
def test_textcnn_random_shapes(case, conv_mode):
    rng = np.random.default_rng(1000 + case)
    _case(seed=case, **_random_shape(rng))


@pytest.mark.parametrize("kind", ["global", "local"])
def test_padding_runs_are_encoded_once_and_exactly(kind):
    """RBR_CONV_PAD_RUNS (D-ATT's un-masked convs over right-padded documents, dual_att/layers.py:43-53,81-89): 32-token slabs of
    pure padding behind another such slab are not computed.  Features AND first argmax are the bits of the full computation --
    also with a non-zero padding row (a pretrained table) and with documents that are all padding or have none."""
    from review_based_recommender_amd import _lib, functional as RF
    _lib.lib().rbr_set_conv_mode(2)
    try:
        g = torch.Generator().manual_seed(11)
        n_docs, L, E, V = 64, 512, 100, 3000
        table = torch.randn(V, E, generator=g).to(DEV)             # row 0 (padding) NOT zero
        ids = torch.randint(1, V, (n_docs, L), generator=g)
        lens = torch.randint(0, L + 1, (n_docs,), generator=g)
        lens[0], lens[1], lens[2], lens[3] = 0, L, 1, L - 1
        ids[torch.arange(L)[None, :] >= lens[:, None]] = 0
        ids = ids.to(DEV)
        if kind == "global":
            ws = [(torch.randn(40, E, k, generator=g) * 0.05).to(DEV) for k in (2, 3, 4)]
            gate = torch.sigmoid(torch.randn(n_docs, 1, generator=g)).expand(n_docs, L).contiguous().to(DEV)    # one scalar per document
            kw = dict(pad_mode=RF.PAD_VALID)
        else:
            ws = [(torch.randn(96, E, 1, generator=g) * 0.05).to(DEV)]
            wg = (torch.randn(1, E, 5, generator=g) * 0.1).to(DEV)
            gate = RF.datt_gate(table, wg, torch.zeros(1, device=DEV), ids, is_global=False, padding_idx=0)      # a function of 5 tokens
            kw = dict(pad_mode=RF.PAD_SAME)
        bs = [(torch.randn(w.shape[0], generator=g) * 0.1).to(DEV) for w in ws]
        outs = []
        for runs in (False, True):
            with torch.no_grad():
                f, am = RF.textcnn(table, ids, None, ws, bs, gate=gate, act=RF.ACT_TANH, padding_idx=0, return_argmax=True,
                                   pad_runs=runs, **kw)
            outs.append((f.clone(), am.clone()))
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][1], outs[1][1])
    finally:
        _lib.lib().rbr_set_conv_mode(0)


@pytest.mark.parametrize("runs", [False, True])
def test_split_gate_conv_equals_the_two_gated_convs(runs):
    """RBR_CONV_GATE_SPLIT: D-ATT's two gated convs of a tower (1-wide local conv under the per-token gate, 2/3/4-wide global convs
    under the per-document gate; dual_att/layers.py:43-53,81-89) as ONE four-bank conv call.  Features and first argmax are the
    bits of the two separate calls; every gradient (table, both gates, weights, biases) agrees to summation order."""
    from review_based_recommender_amd import _lib, functional as RF
    _lib.lib().rbr_set_conv_mode(2)
    try:
        g = torch.Generator().manual_seed(21)
        n_docs, L, E, V = 48, 160, 100, 900
        ids = torch.randint(1, V, (n_docs, L), generator=g)
        lens = torch.randint(0, L + 1, (n_docs,), generator=g)
        lens[0], lens[1], lens[2] = 0, L, 3
        ids[torch.arange(L)[None, :] >= lens[:, None]] = 0
        ids = ids.to(DEV)
        wl = (torch.randn(1, E, 5, generator=g) * 0.1).to(DEV)
        wgl = (torch.randn(1, E, L, generator=g) * 0.02).to(DEV)

        def leaves():
            gg = torch.Generator().manual_seed(22)
            table = torch.randn(V, E, generator=gg).to(DEV).requires_grad_(True)
            ws = [(torch.randn(56, E, 1, generator=gg) * 0.05).to(DEV).requires_grad_(True)] + \
                 [(torch.randn(24, E, k, generator=gg) * 0.05).to(DEV).requires_grad_(True) for k in (2, 3, 4)]
            bs = [(torch.randn(w.shape[0], generator=gg) * 0.1).to(DEV).requires_grad_(True) for w in ws]
            return table, ws, bs

        def gates(table):
            ga = RF.datt_gate(table, wl, torch.zeros(1, device=DEV), ids, is_global=False, padding_idx=0)
            gb = RF.datt_gate(table, wgl, torch.zeros(1, device=DEV), ids, is_global=True, padding_idx=0)
            return ga, gb

        d_out = torch.randn(n_docs, 56 + 72, generator=g).to(DEV)
        # merged
        t1, ws1, bs1 = leaves()
        ga, gb = gates(t1.detach())
        ga, gb = ga.detach().requires_grad_(True), gb.detach().requires_grad_(True)
        f1, am1 = RF.textcnn(t1, ids, None, ws1, bs1, gate=(ga, gb), gate_split=1, pad_mode=RF.PAD_VALID, act=RF.ACT_TANH,
                             padding_idx=0, return_argmax=True, pad_runs=runs)
        (f1 * d_out).sum().backward()
        # separate
        t2, ws2, bs2 = leaves()
        ha, hb = ga.detach().clone().requires_grad_(True), gb.detach().clone().requires_grad_(True)
        fa, ama = RF.textcnn(t2, ids, None, ws2[:1], bs2[:1], gate=ha, pad_mode=RF.PAD_SAME, act=RF.ACT_TANH, padding_idx=0,
                             return_argmax=True, pad_runs=runs)
        fb, amb = RF.textcnn(t2, ids, None, ws2[1:], bs2[1:], gate=hb, pad_mode=RF.PAD_VALID, act=RF.ACT_TANH, padding_idx=0,
                             return_argmax=True, pad_runs=runs)
        f2 = torch.cat([fa, fb], 1)
        (f2 * d_out).sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(f1, f2)
        assert torch.equal(am1, torch.cat([ama, amb], 1))

        def close(a, b, what):
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (what, float((a - b).abs().max()), scale)
        close(t1.grad, t2.grad, "table")
        close(ga.grad, ha.grad, "local gate")
        close(gb.grad, hb.grad, "global gate")
        for k in range(4):
            close(ws1[k].grad, ws2[k].grad, f"W{k}")
            close(bs1[k].grad, bs2[k].grad, f"b{k}")
        # the dense formulation refuses a split gate instead of ignoring it
        _lib.lib().rbr_set_conv_mode(1)
        with pytest.raises(RuntimeError):
            with torch.no_grad():
                RF.textcnn(t1, ids, None, ws1, bs1, gate=(ga.detach(), gb.detach()), gate_split=1, pad_mode=RF.PAD_VALID,
                           act=RF.ACT_TANH, padding_idx=0)
    finally:
        _lib.lib().rbr_set_conv_mode(0)


def test_rows_stationary_gemm_gives_the_ring_kernels_bits(tmp_path):
    """prod_gemm_b16k_kernel (token rows stationary in registers, transposed accumulators) against prod_gemm_b16_kernel (the
    4-stage ring) on the same input: the same plane products into the same accumulators in the same order, so the pooled
    features must agree bit for bit.  The switch is read once per process: two child processes."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from review_based_recommender_amd import functional as RF, _lib\n"
        "_lib.lib().rbr_set_conv_mode(2)\n"
        "g = torch.Generator().manual_seed(5)\n"
        "V, D, n_docs, L = 700, 100, 16, 256\n"
        "table = torch.randn(V, D, generator=g).cuda()\n"
        "ids = torch.randint(0, V, (n_docs, L), generator=g).cuda()\n"
        "ws = [(torch.randn(c, D, k, generator=g) * 0.05).cuda() for k, c in ((1, 200), (2, 100), (3, 100), (4, 100))]\n"
        "bs = [torch.zeros(w.shape[0]).cuda() for w in ws]\n"
        "out = RF.textcnn(table, ids, None, ws, bs, pad_mode=RF.PAD_VALID, act=RF.ACT_TANH, padding_idx=None)\n"
        "np.save(sys.argv[1], out.cpu().numpy())\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for flag in ("1", "0"):
        path = str(tmp_path / ("feat%s.npy" % flag))
        env = dict(os.environ, RBR_GEMM_ROWS_STATIONARY=flag)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
        outs.append(np.load(path))
    assert outs[0].shape == (16, 500) and np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])
