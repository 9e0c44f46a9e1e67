"""Seeded random-shape sweeps of the smaller HIP ops against plain torch CPU restatements (oracle/ref_cpu.py and inline
formulas from the reference files): forward and every gradient, through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import max_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _leaf(t):
    return t.clone().requires_grad_(True)


def _cmp_grads(gpu_leaves, cpu_leaves, rtol=2e-4, atol=2e-6):
    for a, b in zip(gpu_leaves, cpu_leaves):
        scale = float(b.grad.norm()) + 1e-6
        assert max_err(a.grad.cpu().numpy(), b.grad.numpy()) <= atol + rtol * scale


@pytest.mark.parametrize("case", range(12))
def test_datt_gates_random(case, conv_mode):
    """LocalAttention / GlobalAttention gate branches (dual_att/layers.py:34-36,50 and 65-67,84) incl. the token-product gate."""
    from review_based_recommender_amd import functional as RF
    rng = np.random.default_rng(200 + case)
    B, L = int(rng.choice([1, 3, 8])), int(rng.integers(5, 90))
    E, V = int(rng.choice([4, 12, 100])), int(rng.choice([6, 40, 300]))
    win = int(rng.choice([1, 3, 5, 7]))
    is_global = bool(case % 2)
    g = torch.Generator().manual_seed(case)
    table = torch.randn(V, E, generator=g) * 0.5
    ids = torch.randint(0, V, (B, L), generator=g)
    kw = L if is_global else win
    w = torch.randn(1, E, kw, generator=g) / np.sqrt(E * kw)
    b0 = torch.randn(1, generator=g) * 0.1
    cl = [_leaf(table), _leaf(w), _leaf(b0)]
    x = F.embedding(ids, cl[0], padding_idx=0).permute(0, 2, 1)
    ref = torch.sigmoid(F.conv1d(x, cl[1], cl[2], padding=0 if is_global else (win - 1) // 2))      # [B,1,L] or [B,1,1]
    ref = ref.view(B, -1).expand(B, L)
    d = torch.randn(B, L, generator=g)
    (ref * d).sum().backward()
    gl = [_leaf(t.detach().to(DEV)) for t in (table, w, b0)]
    out = RF.datt_gate(gl[0], gl[1], gl[2], ids.to(DEV), is_global=is_global, padding_idx=0)
    (out * d.to(DEV)).sum().backward()
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 2e-6
    _cmp_grads(gl, cl)


@pytest.mark.parametrize("case", range(8))
def test_review_attention_random(case):
    """NARRE LinearAttention (narre.py:40-64)."""
    from oracle import ref_cpu as O
    from review_based_recommender_amd import functional as RF
    rng = np.random.default_rng(300 + case)
    B, R, H, A, NI = int(rng.choice([1, 4, 33])), int(rng.integers(1, 13)), int(rng.choice([5, 24, 150])), int(rng.choice([3, 8, 32])), 17
    g = torch.Generator().manual_seed(case)
    feat = torch.randn(B, R, H, generator=g) * 0.5
    oid = torch.randint(0, NI, (B, R), generator=g)
    ps = [torch.randn(H, A, generator=g) * 0.1, torch.randn(A, A, generator=g) * 0.1, torch.randn(A, 1, generator=g) * 0.1,
          torch.randn(A, generator=g) * 0.1, torch.randn(1, generator=g) * 0.1, torch.randn(NI, A, generator=g) * 0.1]
    cl = [_leaf(feat)] + [_leaf(p) for p in ps]
    ro, ra = O.linear_attention(cl[0], oid, *cl[1:])
    # odd cases: the dropout multiplier that follows the pooled feature (narre.py:62) is applied inside the kernels
    drop = ((torch.rand(B, H, generator=g) > 0.3).float() / 0.7) if case % 2 else None
    if drop is not None:
        ro = ro * drop
    d = torch.randn(B, H, generator=g)
    (ro * d).sum().backward()
    gl = [_leaf(t.detach().to(DEV)) for t in [feat] + ps]
    go, ga = RF.review_attention(gl[0], oid.to(DEV), *gl[1:], pad_idx=0, drop=None if drop is None else drop.to(DEV))
    (go * d.to(DEV)).sum().backward()
    assert max_err(go.detach().cpu().numpy(), ro.detach().numpy()) <= 1e-5
    assert max_err(ga.detach().cpu().numpy(), ra.detach().numpy()) <= 1e-6
    _cmp_grads(gl, cl, atol=1e-5)      # R = 1 makes d(logit) = att * (d att - sum att * d att) an exact-zero cancellation: rounding noise of the dot order


@pytest.mark.parametrize("case", range(8))
def test_siamese_ops_random(case):
    """review_bag (lookup + variational dropout multiplier + masked mean) and additive attention (+ node dropout multiplier)."""
    from oracle import ref_cpu as O
    from review_based_recommender_amd import functional as RF
    rng = np.random.default_rng(400 + case)
    n_rev, T, D, V = int(rng.choice([1, 5, 40])), int(rng.integers(1, 40)), int(rng.choice([3, 8, 108])), int(rng.choice([5, 60]))
    g = torch.Generator().manual_seed(case)
    table = torch.randn(V, D, generator=g)
    ids = torch.randint(0, V, (n_rev, T), generator=g)
    mask = torch.rand(n_rev, T, generator=g) > 0.3
    drop = (torch.rand(n_rev, D, generator=g) > 0.2).float() / 0.8
    ct = _leaf(table)
    x = F.embedding(ids, ct, padding_idx=0) * drop.unsqueeze(1)
    ref = O.masked_avg_pool(x, mask)
    d = torch.randn(n_rev, D, generator=g)
    (ref * d).sum().backward()
    gt = _leaf(table.to(DEV))
    out = RF.review_bag(gt, ids.to(DEV), mask.to(DEV), drop=drop.to(DEV), padding_idx=0)
    (out * d.to(DEV)).sum().backward()
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    _cmp_grads([gt], [ct])
    # the sorted-run form of the backward (RBR_BAG_BWD=sort) against the same reference
    import os
    os.environ["RBR_BAG_BWD"] = "sort"
    try:
        gt2 = _leaf(table.to(DEV))
        (RF.review_bag(gt2, ids.to(DEV), mask.to(DEV), drop=drop.to(DEV), padding_idx=0) * d.to(DEV)).sum().backward()
    finally:
        del os.environ["RBR_BAG_BWD"]
    _cmp_grads([gt2], [ct])

    B, R, H, K = int(rng.choice([1, 6])), int(rng.integers(1, 12)), int(rng.choice([4, 27, 108])), int(rng.choice([2, 32]))
    rev = torch.randn(B, R, H, generator=g) * 0.5
    rmask = torch.rand(B, R, generator=g) > 0.3
    nd = (torch.rand(B, R, generator=g) > 0.25).float() / 0.75
    ps = [torch.randn(K, H, generator=g) * 0.2, torch.randn(K, generator=g) * 0.1, torch.randn(1, K, generator=g) * 0.3]
    cl = [_leaf(rev)] + [_leaf(p) for p in ps]
    ro, rs = O.additive_attention(cl[0] * nd.unsqueeze(2), rmask, *cl[1:])
    d2 = torch.randn(B, H, generator=g)
    (ro * d2).sum().backward()
    gl = [_leaf(t.to(DEV)) for t in [rev] + ps]
    go, gs = RF.additive_attention(gl[0], rmask.to(DEV), gl[1], gl[2], gl[3], node_drop=nd.to(DEV))
    (go * d2.to(DEV)).sum().backward()
    assert max_err(go.detach().cpu().numpy(), ro.detach().numpy()) <= 1e-5
    assert max_err(gs.detach().cpu().numpy(), rs.detach().numpy()) <= 1e-6
    _cmp_grads(gl, cl)


@pytest.mark.parametrize("case", range(6))
def test_linear_and_head_random(case):
    """nn.Linear with ReLU / Tanh / dropout multiplier on the MFMA GEMM, and LastFeat x2 + FM."""
    from oracle import ref_cpu as O
    from review_based_recommender_amd import functional as RF
    rng = np.random.default_rng(500 + case)
    N, IN, OUT = int(rng.choice([1, 37, 130])), int(rng.choice([3, 50, 129])), int(rng.choice([1, 31, 70]))
    g = torch.Generator().manual_seed(case)
    x, W, b = torch.randn(N, IN, generator=g), torch.randn(OUT, IN, generator=g) / np.sqrt(IN), torch.randn(OUT, generator=g) * 0.1
    drop = (torch.rand(N, OUT, generator=g) > 0.5).float() * 2
    act = ["none", "relu", "tanh"][case % 3]
    cl = [_leaf(t) for t in (x, W, b)]
    z = F.linear(*cl)
    z = F.relu(z) if act == "relu" else torch.tanh(z) if act == "tanh" else z
    ref = z * drop
    d = torch.randn(N, OUT, generator=g)
    (ref * d).sum().backward()
    gl = [_leaf(t.to(DEV)) for t in (x, W, b)]
    out = RF.linear(gl[0], gl[1], gl[2], relu=(act == "relu"), tanh=(act == "tanh"), drop=drop.to(DEV))
    (out * d.to(DEV)).sum().backward()
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 2e-5
    _cmp_grads(gl, cl)

    B, H, K, U, I = int(rng.choice([1, 9, 64])), int(rng.choice([4, 150])), int(rng.choice([3, 32, 40])), 11, 13
    uf, itf = torch.randn(B, H, generator=g) * 0.3, torch.randn(B, H, generator=g) * 0.3
    uid, iid = torch.randint(0, U, (B,), generator=g), torch.randint(0, I, (B,), generator=g)
    ps = [torch.randn(H, K, generator=g) * 0.1, torch.randn(K, generator=g) * 0.1, torch.randn(U, K, generator=g) * 0.1,
          torch.randn(H, K, generator=g) * 0.1, torch.randn(K, generator=g) * 0.1, torch.randn(I, K, generator=g) * 0.1,
          torch.randn(K, 1, generator=g) * 0.3, torch.randn(1, generator=g), torch.randn(U, 1, generator=g) * 0.1,
          torch.randn(I, 1, generator=g) * 0.1]
    cl = [_leaf(uf), _leaf(itf)] + [_leaf(p) for p in ps]
    ul = O.last_feat(cl[0], uid, cl[2], cl[3], cl[4])
    il = O.last_feat(cl[1], iid, cl[5], cl[6], cl[7])
    ref = O.fm(ul, il, uid, iid, cl[8], cl[9], cl[10], cl[11]).view(-1)
    d3 = torch.randn(B, generator=g)
    (ref * d3).sum().backward()
    gl = [_leaf(t.to(DEV)) for t in [uf, itf] + ps]
    out = RF.pair_head(gl[0], gl[1], uid.to(DEV), iid.to(DEV), *gl[2:], pad_u=0, pad_i=0)
    (out * d3.to(DEV)).sum().backward()
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    _cmp_grads(gl, cl)


@pytest.mark.parametrize("N,IN,OUT,act", [(1024, 500, 500, "relu"), (1024, 500, 50, "none"), (300, 1000, 70, "tanh"), (64, 129, 33, "relu"), (64, 50, 33, "relu")])
def test_linear_split_along_k_matches_torch_and_is_reproducible(N, IN, OUT, act):
    """rbr_linear_fwd_ex / _bwd_ex (D-ATT's shared fc, dual_att.py:31-35, at its cfg4 shapes and two odd ones): products with few
    output tiles are split along K, the last slice of a tile adds the partial tiles in slice order.  Against torch in float64;
    two runs give the same bits (the partial tiles are added in slice order)."""
    from review_based_recommender_amd import _lib, functional as RF
    g = torch.Generator().manual_seed(N + OUT)
    x, W, b = torch.randn(N, IN, generator=g), torch.randn(OUT, IN, generator=g) / np.sqrt(IN), torch.randn(OUT, generator=g) * 0.1
    drop = (torch.rand(N, OUT, generator=g) > 0.5).float() * 2
    d = torch.randn(N, OUT, generator=g)
    cl = [_leaf(t.double()) for t in (x, W, b)]
    z = F.linear(*cl)
    z = F.relu(z) if act == "relu" else torch.tanh(z) if act == "tanh" else z
    ((z * drop.double()) * d.double()).sum().backward()
    L_ = _lib.lib()
    split = L_.rbr_linear_fwd_ws_floats(N, IN, OUT) > 0
    assert split == (IN >= 128), "K of at least four 32-deep chunks is split (every shape here has few output tiles), a shorter K not"
    outs = []
    for _ in range(2):
        gl = [_leaf(t.to(DEV)) for t in (x, W, b)]
        out = RF.linear(gl[0], gl[1], gl[2], relu=(act == "relu"), tanh=(act == "tanh"), drop=drop.to(DEV))
        (out * d.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        outs.append([out.detach().clone()] + [t.grad.clone() for t in gl])
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    ref = [(z * drop.double()).detach()] + [t.grad for t in cl]
    for got, r in zip(outs[0], ref):
        scale = float(r.abs().max()) + 1e-30
        assert float((got.double().cpu() - r).abs().max()) <= 2e-5 * max(1.0, scale), (tuple(r.shape), scale)


@pytest.mark.parametrize("want_dx,want_db", [(True, True), (False, True), (True, False), (False, False)])
def test_linear_backward_ex_without_input_or_bias_gradient(want_dx, want_db):
    """rbr_linear_bwd_ex through the C ABI with d_x and / or db NULL: the weight gradient alone is one product (gemm_kernel), with the
    input gradient both are one launch (gemm2_kernel); their reductions and the bias column sums share the second launch
    (gemm_reduce2_kernel).  Every combination against torch in float64; what was not asked for is not written."""
    from review_based_recommender_amd import _lib
    from review_based_recommender_amd._lib import dev_ptr
    L_ = _lib.lib()
    N, IN, OUT = 1024, 500, 50                                   # D-ATT's second fc layer: dW split along K, d_x not
    g = torch.Generator().manual_seed(3)
    x, W = torch.randn(N, IN, generator=g), torch.randn(OUT, IN, generator=g) / np.sqrt(IN)
    d_y = torch.randn(N, OUT, generator=g)
    y = x @ W.t()
    xd, Wd, yd, dyd = (t.to(DEV) for t in (x, W, y, d_y))
    F32 = torch.float32
    dW = torch.full((OUT, IN), 7.0, device=DEV)
    d_x = torch.full((N, IN), 7.0, device=DEV)
    db = torch.full((OUT,), 7.0, device=DEV)
    ws = torch.empty(L_.rbr_linear_bwd_ex_ws_floats(N, IN, OUT), device=DEV)
    rc = L_.rbr_linear_bwd_ex(N, IN, OUT, dev_ptr(xd, F32, "x"), dev_ptr(Wd, F32, "W"), dev_ptr(yd, F32, "y"), dev_ptr(dyd, F32, "d_y"), 0,
                              None, dev_ptr(d_x, F32, "d_x") if want_dx else None, dev_ptr(dW, F32, "dW"),
                              dev_ptr(db, F32, "db") if want_db else None, dev_ptr(ws, F32, "ws"), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L_.rbr_last_error()
    torch.cuda.synchronize()
    ref_dW = d_y.double().t() @ x.double()
    assert float((dW.double().cpu() - ref_dW).abs().max()) <= 2e-5 * float(ref_dW.abs().max())
    if want_dx:
        ref_dx = d_y.double() @ W.double()
        assert float((d_x.double().cpu() - ref_dx).abs().max()) <= 2e-5 * float(ref_dx.abs().max())
    else:
        assert float((d_x - 7.0).abs().max()) == 0.0
    if want_db:
        assert float((db.double().cpu() - d_y.double().sum(0)).abs().max()) <= 1e-4
    else:
        assert float((db - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("n", [1, 7, 256, 1000, 70001])
def test_mse_loss_matches_torch(n):
    """rbr_mse_loss_fwd/_bwd vs nn.MSELoss (train_deepconn_pp.py:137,164): loss, d_pred, and a non-unit upstream gradient."""
    from review_based_recommender_amd import functional as RF
    g = torch.Generator().manual_seed(n)
    pred, target = torch.randn(n, generator=g) * 2 + 3, torch.rand(n, generator=g) * 4 + 1
    pc = _leaf(pred)
    ref = F.mse_loss(pc, target)
    (ref * 0.37).backward()
    pg = _leaf(pred.to(DEV))
    out = RF.mse_loss(pg, target.to(DEV))
    (out * 0.37).backward()
    assert out.shape == ref.shape
    assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref))
    assert max_err(pg.grad.cpu().numpy(), pc.grad.numpy()) <= 1e-6 * float(pc.grad.abs().max()) + 1e-12
    with pytest.raises(RuntimeError):
        RF.mse_loss(pg, target.to(DEV)[: max(n - 1, 0)])


def test_stack_rows_is_a_view_for_adjacent_inputs():
    """stack_rows == torch.cat; no copy for the two halves of one block (clone_adjacent), a copy otherwise."""
    from review_based_recommender_amd import functional as RF
    g = torch.Generator().manual_seed(0)
    a, b = torch.randint(0, 99, (5, 7), generator=g).to(DEV), torch.randint(0, 99, (5, 7), generator=g).to(DEV)
    m = torch.rand(5, generator=g).to(DEV)
    assert torch.equal(RF.stack_rows(a, b), torch.cat([a, b]))
    ca, cb, cm = RF.clone_adjacent((a, b, m))
    assert torch.equal(ca, a) and torch.equal(cb, b) and torch.equal(cm, m)
    st = RF.stack_rows(ca, cb)
    assert st.data_ptr() == ca.data_ptr() and torch.equal(st, torch.cat([a, b]))
    assert RF.stack_rows(cb, ca).data_ptr() not in (ca.data_ptr(), cb.data_ptr())          # wrong order: a real cat
    ra = ca.view(5, 7)[:, :]            # reshaped views of the halves keep the adjacency (NARRE's [bz*R, T] view)
    assert RF.stack_rows(ra.reshape(-1, 7), cb.reshape(-1, 7)).data_ptr() == ca.data_ptr()


@pytest.mark.parametrize("case", range(4))
def test_pair_head_stacked_equals_separate(case):
    """pair_head on the encoder's [2B,H] output == pair_head on its two halves: prediction and every gradient, bit for bit
    on the dense ones (same kernels, same order)."""
    from review_based_recommender_amd import functional as RF
    rng = np.random.default_rng(900 + case)
    B, H, K, NU, NI = int(rng.choice([1, 5, 64])), int(rng.choice([6, 150])), int(rng.choice([4, 32])), 23, 31
    g = torch.Generator().manual_seed(case)
    feat = torch.randn(2 * B, H, generator=g).to(DEV)
    uid, iid = torch.randint(0, NU, (B,), generator=g).to(DEV), torch.randint(0, NI, (B,), generator=g).to(DEV)
    shapes = [(H, K), (K,), (NU, K), (H, K), (K,), (NI, K), (K, 1), (1,), (NU, 1), (NI, 1)]
    params = [(torch.randn(*s, generator=g) * 0.3).to(DEV) for s in shapes]
    drop = (torch.rand(B, K, generator=g) > 0.3).float().div(0.7).to(DEV)
    d = torch.randn(B, generator=g).to(DEV)
    res = []
    for stacked in (True, False):
        f = _leaf(feat)
        ps = [_leaf(t) for t in params]
        pred = RF.pair_head(f, None, uid, iid, *ps, drop=drop) if stacked else RF.pair_head(f[:B], f[B:], uid, iid, *ps, drop=drop)
        (pred * d).sum().backward()
        res.append((pred.detach(), f.grad, [t.grad for t in ps]))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert max_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6 * (float(b.abs().max()) + 1e-6)      # embedding rows: atomics


def test_dropout_multiplier_statistics_and_replay():
    """rbr_dropout_multiplier: values in {0, 1/(1-p)}, drop rate ~ p, a new mask per call (device-side call counter), also on
    every replay of a captured graph; the same seed and call number reproduce the same mask."""
    from review_based_recommender_amd import functional as RF
    dev = torch.device(DEV)
    for p in (0.1, 0.5, 0.8):
        m = RF.dropout_multiplier((1000, 333), p, True, dev)
        vals = torch.unique(m)
        assert set(np.round(vals.cpu().numpy(), 5)) == {0.0, np.round(np.float32(1.0 / (1.0 - p)), 5)}
        rate = float((m == 0).float().mean())
        assert abs(rate - p) < 4 * np.sqrt(p * (1 - p) / m.numel()) + 1e-3
        assert abs(float(m.mean()) - 1.0) < 0.02            # E[multiplier] = 1
    assert RF.dropout_multiplier((4, 4), 0.5, False, dev) is None and RF.dropout_multiplier((4, 4), 0.0, True, dev) is None
    assert float(RF.dropout_multiplier((4, 4), 1.0, True, dev).abs().sum()) == 0.0
    a, b = RF.dropout_multiplier((64, 32), 0.5, True, dev), RF.dropout_multiplier((64, 32), 0.5, True, dev)
    assert not torch.equal(a, b)
    # same (seed, call number) -> same mask: rewind the device counter
    state = RF._DROP_STATE[(dev.index, 0)]
    torch.cuda.synchronize()
    call = int(state[0])
    c1 = RF.dropout_multiplier((777,), 0.3, True, dev)
    state[0] = call
    c2 = RF.dropout_multiplier((777,), 0.3, True, dev)
    assert torch.equal(c1, c2) and int(state[0]) == call + 1 and int(state[1]) == 0
    # a captured launch draws a new mask on every replay
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        RF.dropout_multiplier((256, 32), 0.5, True, dev)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = RF.dropout_multiplier((256, 32), 0.5, True, dev)
    masks = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        masks.append(out.clone())
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])


@pytest.mark.parametrize("case", range(4))
def test_pair_head_training_forward_draws_the_same_mask(case):
    """pair_head(drop=p) (one launch: mask drawn in the kernel, gradient buffer cleared by spare workgroups) == dropout_multiplier
    + pair_head(drop=mask) for the same call number: prediction and every gradient; the call number advances by one."""
    from review_based_recommender_amd import functional as RF
    rng = np.random.default_rng(950 + case)
    B, H, K, NU, NI = int(rng.choice([1, 7, 64, 300])), int(rng.choice([6, 150])), int(rng.choice([3, 32, 40])), 23, 31
    p = [0.0, 0.3, 0.5, 0.9][case]
    g = torch.Generator().manual_seed(case)
    feat = torch.randn(2 * B, H, generator=g).to(DEV)
    uid, iid = torch.randint(0, NU, (B,), generator=g).to(DEV), torch.randint(0, NI, (B,), generator=g).to(DEV)
    shapes = [(H, K), (K,), (NU, K), (H, K), (K,), (NI, K), (K, 1), (1,), (NU, 1), (NI, 1)]
    params = [(torch.randn(*s, generator=g) * 0.3).to(DEV) for s in shapes]
    d = torch.randn(B, generator=g).to(DEV)
    _, state = RF._drop_rng(torch.device(DEV))
    torch.cuda.synchronize()
    call = int(state[0])
    res = []
    for fused in (True, False):
        state[0] = call
        f = _leaf(feat)
        ps = [_leaf(t) for t in params]
        drop = float(p) if fused else RF.dropout_multiplier((B, K), p, True, torch.device(DEV))
        pred = RF.pair_head(f, None, uid, iid, *ps, drop=drop, pad_u=0, pad_i=0)
        (pred * d).sum().backward()
        torch.cuda.synchronize()
        assert int(state[0]) == call + (1 if p > 0 else 0) and int(state[1]) == 0
        res.append((pred.detach(), f.grad, [t.grad for t in ps]))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert max_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6 * (float(b.abs().max()) + 1e-6)


def test_mse_unit_gradient_shortcut():
    """loss.backward(unit_scalar) returns the gradient the forward launch already wrote; == the launch path and torch."""
    from review_based_recommender_amd import functional as RF
    g = torch.Generator().manual_seed(5)
    pred, target = torch.randn(300, generator=g), torch.randn(300, generator=g)
    pc = _leaf(pred)
    F.mse_loss(pc, target).backward()
    a, b = _leaf(pred.to(DEV)), _leaf(pred.to(DEV))
    RF.mse_loss(a, target.to(DEV)).backward(RF.unit_scalar(torch.device(DEV)))
    RF.mse_loss(b, target.to(DEV)).backward()
    assert max_err(a.grad.cpu().numpy(), pc.grad.numpy()) <= 1e-7 and max_err(b.grad.cpu().numpy(), pc.grad.numpy()) <= 1e-7
