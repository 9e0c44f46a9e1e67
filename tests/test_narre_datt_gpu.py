"""NARRE and D-ATT on the HIP path vs golden vectors captured from the reference (through the C ABI)."""
import pytest
import torch

import synth
from helpers import check_grads, check_params_after, golden, max_err, quiet

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-4
DEV = "cuda:0"


def _narre(cfg, dropout=0.0):
    from review_based_recommender_amd.models.narre.narre import NARRE
    c = cfg
    m = quiet(NARRE, c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"], c["R"], c["T"], dropout, 0, 0, 0,
              None, "CNN")
    m.load_state_dict(synth.narre_params(cfg, 0))
    return m.to(DEV)


def _narre_batch(b):
    keys = ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid")
    return tuple(b[k].to(DEV) for k in keys), b["ratings"].to(DEV)


@pytest.mark.parametrize("name,cfgname,edge", [("narre_tiny", "tiny", True), ("narre_small", "small", True),
                                               ("narre_cfg3", "cfg3", False)])
def test_narre_matches_reference(golden_dir, name, cfgname, edge, conv_mode):
    from review_based_recommender_amd.train_step import make_optimizer, train_step
    g = golden(golden_dir, name)
    cfg = synth.NARRE_CFGS[cfgname]
    model = _narre(cfg)
    args, ratings = _narre_batch(synth.narre_batch(cfg, 1, edge_cases=edge))
    model.eval()
    with torch.no_grad():
        pred, ua, ia = model(*args)
    assert pred.shape == (cfg["B"],) and ua.shape == (cfg["B"], cfg["R"], 1)
    assert max_err(pred.cpu().numpy(), g["pred_eval"]) <= FWD_TOL
    assert max_err(ua.cpu().numpy(), g["u_att"]) <= 1e-5
    assert max_err(ia.cpu().numpy(), g["i_att"]) <= 1e-5

    model.train()
    loss = torch.nn.functional.mse_loss(model(*args)[0], ratings)
    loss.backward()
    check_grads({k: p.grad for k, p in model.named_parameters()}, g)
    model.zero_grad()
    opt = make_optimizer(model)
    for step in range(3):
        loss, gnorm, pred = train_step(model, opt, args, ratings)
        if step == 0:
            assert abs(float(loss) - float(g["loss"])) <= 1e-4
            assert abs(float(gnorm) - float(g["gnorm"])) <= 2e-4 * float(g["gnorm"])
        if step in (0, 2):
            check_params_after(model, g, f"after{step + 1}")


def test_narre_state_dict_keys():
    cfg = synth.NARRE_CFGS["tiny"]
    sd = synth.narre_params(cfg, 0)
    m = _narre(cfg)
    assert list(m.state_dict().keys()) == list(sd.keys())


def _datt(cfg, dropout=0.0, scale=1.0):
    from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
    c = cfg
    m = quiet(DualAtt, c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], dropout, None)
    m.load_state_dict(synth.datt_params(cfg, 0, table_scale=scale))
    return m.to(DEV)


@pytest.mark.parametrize("name,cfgname", [("datt_tiny", "tiny"), ("datt_small", "small")])
def test_datt_matches_reference(golden_dir, name, cfgname, conv_mode):
    from review_based_recommender_amd.train_step import make_optimizer, train_step
    g = golden(golden_dir, name)
    cfg = synth.DATT_CFGS[cfgname]
    model = _datt(cfg)
    b = synth.datt_batch(cfg, 1, edge_cases=True)
    args, ratings = (b["u_docs"].to(DEV), b["i_docs"].to(DEV)), b["ratings"].to(DEV)
    model.eval()
    with torch.no_grad():
        pred = model(*args)
    assert max_err(pred.cpu().numpy(), g["pred_eval"]) <= FWD_TOL

    model.train()
    loss = torch.nn.functional.mse_loss(model(*args), ratings)
    loss.backward()
    check_grads({k: p.grad for k, p in model.named_parameters()}, g)
    model.zero_grad()
    opt = make_optimizer(model)
    for step in range(3):
        loss, gnorm, pred = train_step(model, opt, args, ratings)
        if step == 0:
            assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(1.0, float(g["loss"]))
        if step in (0, 2):
            check_params_after(model, g, f"after{step + 1}")


def test_datt_cfg4_full_step(golden_dir, conv_mode):
    """BASELINE configs[3]: B=512, 2x1024 tokens, E=100 -- eval predictions, then the trainer step against the reference's
    loss, every gradient (gated global convs and both gates included: models/dual_att/layers.py:43-53,81-89), the clipped
    norm and the parameters after 1 and 3 Adam steps (fixture: tests/golden/make_golden.py:gen_datt, big=True)."""
    from review_based_recommender_amd.train_step import make_optimizer, train_step
    g = golden(golden_dir, "datt_cfg4")
    cfg = synth.DATT_CFGS["cfg4"]
    model = _datt(cfg, scale=0.3)
    b = synth.datt_batch(cfg, 1)
    args, ratings = (b["u_docs"].to(DEV), b["i_docs"].to(DEV)), b["ratings"].to(DEV)
    model.eval()
    with torch.no_grad():
        pred = model(*args)
    assert max_err(pred.cpu().numpy(), g["pred_eval"]) <= FWD_TOL

    model.train()
    pred = model(*args)
    assert max_err(pred.detach().cpu().numpy(), g["pred"]) <= FWD_TOL
    loss = torch.nn.functional.mse_loss(pred, ratings)
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(1.0, float(g["loss"]))
    loss.backward()
    check_grads({k: p.grad for k, p in model.named_parameters()}, g)
    model.zero_grad()
    opt = make_optimizer(model)
    for step in range(3):
        loss, gnorm, pred = train_step(model, opt, args, ratings)
        if step == 0:
            assert abs(float(gnorm) - float(g["gnorm"])) <= 2e-4 * float(g["gnorm"])
        if step in (0, 2):
            check_params_after(model, g, f"after{step + 1}")


@pytest.mark.parametrize("L", [512, 1280])
def test_global_gate_backward_over_token_rows_matches_the_plain_one(L):
    """rbr_datt_global_gate_bwd_rows (private row-compacted copies of the table gradient, dtable overwritten) against
    rbr_datt_global_gate_bwd (atomics straight into a zeroed dtable): same dw / db0 bit for bit, same dtable to f32 sum
    reordering; Zipf-like ids so that hot rows exist, pad tokens included."""
    from review_based_recommender_amd import functional as RF
    gen = torch.Generator().manual_seed(5)
    B, V, E = 24, 3000, 100            # L = 1280: rows longer than the 1024 positions a wave holds at a time
    ids = (torch.rand(B, L, generator=gen) ** 4 * V).long().clamp_(0, V - 1).to(DEV)
    table = (torch.randn(V, E, generator=gen) * 0.3).to(DEV).requires_grad_()
    w = (torch.randn(1, E, L, generator=gen) * 0.05).to(DEV).requires_grad_()
    b0 = torch.zeros(1, device=DEV, requires_grad=True)
    up = torch.randn(B, L, generator=gen).to(DEV)
    rows = RF.datt_token_rows(ids, V)
    assert rows is not None
    outs = []
    for r in (None, rows):
        gate = RF.datt_gate(table, w, b0, ids, is_global=True, padding_idx=0, rows=r)
        outs.append(torch.autograd.grad((gate * up).sum(), (table, w, b0)))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    scale = float(outs[0][0].abs().max())
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-6 * scale   # hot rows sum thousands of f32 terms in another order
    assert float(outs[1][0][0].abs().max()) == 0.0                       # the pad row gets no gradient
    assert RF.datt_token_rows(ids[:2, :64], V) is None                   # too few positions: the plain path is kept


def test_global_gate_backward_in_two_calls_equals_one():
    """rbr_datt_global_gate_bwd_rows as two calls -- dtable == NULL (dw, db0; dpre stays in ws), then accumulate | 2 (the table
    rows alone, possibly on another stream: functional._DattTowers puts tower 1's there) -- against the single call: the same
    kernels on the same inputs; the second phase reads neither table nor gate nor dgate (NULL)."""
    import ctypes as C
    from review_based_recommender_amd import _lib
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd._lib import dev_ptr
    F32, I64 = torch.float32, torch.int64
    L_ = _lib.lib()
    gen = torch.Generator().manual_seed(9)
    B, L, V, E = 16, 512, 2000, 100
    ids = (torch.rand(B, L, generator=gen) ** 4 * V).long().clamp_(0, V - 1).to(DEV)
    table = (torch.randn(V, E, generator=gen) * 0.3).to(DEV)
    w = (torch.randn(1, E, L, generator=gen) * 0.05).to(DEV)
    gate = torch.sigmoid(torch.randn(B, 1, generator=gen)).expand(B, L).contiguous().to(DEV)
    dgate = torch.randn(B, L, generator=gen).to(DEV)
    rows = RF.datt_token_rows(ids, V)
    assert rows is not None
    st = torch.cuda.current_stream().cuda_stream
    n_ws = L_.rbr_datt_global_gate_bwd_rows_ws_floats(B, L, E, V)

    def call(dtable, flags, full=True, ws=None, dw=None, db0=None):
        return L_.rbr_datt_global_gate_bwd_rows(B, L, E, V, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "t") if full else None,
                                                dev_ptr(w, F32, "w"), dev_ptr(gate, F32, "g") if full else None,
                                                dev_ptr(dgate, F32, "dg") if full else None, 0,
                                                dev_ptr(dw, F32, "dw") if dw is not None else None,
                                                dev_ptr(db0, F32, "db0") if db0 is not None else None,
                                                dev_ptr(dtable, F32, "dt") if dtable is not None else None, dev_ptr(ws, F32, "ws"),
                                                rows.data_ptr(), flags, st)

    base = torch.randn(V, E, generator=gen).to(DEV)       # accumulate mode: rows are added to what the buffer holds
    one, dw1, db1, ws1 = base.clone(), torch.empty_like(w), torch.empty(1, device=DEV), torch.empty(n_ws, device=DEV)
    assert call(one, 1, ws=ws1, dw=dw1, db0=db1) == 0
    two, dw2, db2, ws2 = base.clone(), torch.empty_like(w), torch.empty(1, device=DEV), torch.empty(n_ws, device=DEV)
    assert call(None, 1, ws=ws2, dw=dw2, db0=db2) == 0
    assert torch.equal(two, base)                          # the first phase writes no table row
    assert call(two, 1 | 2, full=False, ws=ws2) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    # (the occurrence matrix is filled with float atomics: a hot row's sum may round differently from run to run)
    assert float((one - two).abs().max()) <= 2e-6 * float((one - base).abs().max())
    assert call(None, 1 | 2, full=False, ws=ws2) != 0      # rows-only without a table gradient buffer: refused
    assert b"null pointer" in L_.rbr_last_error()


def test_datt_shared_table_gradient_buffer_survives_interleaved_steps():
    """functional.table_fanout: the eight producers of word-table gradient of a D-ATT step add their rows into one buffer that
    hangs on the step's own fan-out node.  Two micro-batches run forward, forward, backward, backward (gradient accumulation)
    must give the sum of their separately computed gradients -- a buffer shared ACROSS the two steps would not."""
    from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
    torch.manual_seed(11)
    V, L, B = 2000, 512, 32
    m = quiet(DualAtt, V, L, 5, 40, 24, 100, 64, 16, 0.0, None).to(DEV)
    with torch.no_grad():
        m.word_embeddings.embedding.weight.mul_(0.3)
    gen = torch.Generator().manual_seed(3)
    batches = [tuple((torch.rand(B, L, generator=gen) ** 3 * V).long().clamp_(1, V - 1).to(DEV) for _ in range(2)) for _ in range(2)]
    ys = [torch.randn(B, generator=gen).to(DEV) for _ in range(2)]
    table = m.word_embeddings.embedding.weight
    singles = []
    for b, y in zip(batches, ys):
        m.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(m(*b), y).backward()
        singles.append({k: p.grad.clone() for k, p in m.named_parameters()})
    assert float(singles[0]["word_embeddings.embedding.weight"].abs().max()) > 0
    m.zero_grad(set_to_none=True)
    losses = [torch.nn.functional.mse_loss(m(*b), y) for b, y in zip(batches, ys)]      # forward, forward
    losses[0].backward()
    losses[1].backward()                                                               # backward, backward
    for k, p in m.named_parameters():
        want = singles[0][k] + singles[1][k]
        assert float((p.grad - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), k
    assert table.grad.shape == table.shape


def test_datt_state_dict_keys():
    cfg = synth.DATT_CFGS["tiny"]
    assert list(_datt(cfg).state_dict().keys()) == list(synth.datt_params(cfg, 0).keys())


def test_hierpooling_matches_reference(golden_dir):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    g = golden(golden_dir, "deepconn_hier_small")
    cfg = synth.DEEPCONN_CFGS["small"]
    m = quiet(DeepCoNNpp, cfg["U"], cfg["I"], cfg["V"], cfg["kz"][:1], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.0,
              "HierPooling")
    m.load_state_dict(synth.deepconn_hier_params(cfg, 0))
    m.to(DEV)
    b = synth.deepconn_batch(cfg, 1, edge_cases=True)
    args = tuple(b[k].to(DEV) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids"))
    m.eval()
    with torch.no_grad():
        assert max_err(m(*args).cpu().numpy(), g["pred_eval"]) <= FWD_TOL
    m.train()
    loss = torch.nn.functional.mse_loss(m(*args), b["ratings"].to(DEV))
    loss.backward()
    check_grads({k: p.grad for k, p in m.named_parameters()}, g)


def test_standalone_layers_match_oracle(conv_mode):
    """WordEmbedding.forward and NgramFeat.forward(inputs, masks) with the reference's per-layer signatures."""
    from oracle import ref_cpu as O
    from review_based_recommender_amd.models.deepconn.layers import NgramFeat, WordEmbedding
    cfg = synth.DEEPCONN_CFGS["small"]
    p = synth.deepconn_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1, edge_cases=True)
    we = quiet(WordEmbedding, cfg["V"], cfg["D"]).to(DEV)
    we.embedding.weight.data.copy_(p["word_embeddings.embedding.weight"])
    ng = quiet(NgramFeat, cfg["kz"], cfg["D"], cfg["H"], cfg["L"]).to(DEV)
    ng.load_state_dict({k[len("ngram."):]: v for k, v in p.items() if k.startswith("ngram.")})
    emb = we(b["u_docs"].to(DEV))
    ref_emb = O.word_embedding(p["word_embeddings.embedding.weight"], b["u_docs"])
    assert torch.equal(emb.cpu(), ref_emb)
    emb = emb.detach().requires_grad_(True)
    out = ng(emb, b["u_masks"].to(DEV))
    ws, bs = O.conv_params(p)
    ref_in = ref_emb.clone().requires_grad_(True)
    ref = O.ngram_feat_cnn(ref_in, b["u_masks"], ws, bs)
    assert out.shape == (cfg["B"], cfg["H"], 1)
    assert max_err(out.detach().cpu().numpy()[..., 0], ref.detach().numpy()) <= 2e-5
    out.sum().backward()
    ref.sum().backward()
    assert max_err(emb.grad.cpu().numpy(), ref_in.grad.numpy()) <= 2e-5
