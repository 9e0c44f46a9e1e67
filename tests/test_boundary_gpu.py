"""Drop-in boundary beyond the fused model path: the L1 layers' own forward signatures (reference
models/deepconn/layers.py:46-60 MyConv1d, :156-165 LastFeat, :189-209 FM) against the oracle, and the device-side range
check that stands in for nn.Embedding's IndexError (layers.py:23)."""
import pytest
import torch

import synth
from helpers import max_err, quiet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _leaf(t):
    return t.detach().clone().requires_grad_(True)


def test_myconv1d_forward_matches_oracle():
    from oracle import ref_cpu as O
    from review_based_recommender_amd.models.deepconn.layers import MyConv1d
    g = torch.Generator().manual_seed(3)
    bz, cin, L, cout = 5, 24, 37, 30
    conv = quiet(MyConv1d, "3,5,7", cin, cout).to(DEV)
    x = torch.randn(bz, cin, L, generator=g)
    x_d = _leaf(x.to(DEV))
    out = conv(x_d)
    ws = [_leaf(c.weight.cpu()) for c in conv.list_of_conv1d]
    bs = [_leaf(c.bias.cpu()) for c in conv.list_of_conv1d]
    x_c = _leaf(x)
    ref = O.my_conv1d(x_c, ws, bs)
    assert out.shape == ref.shape == (bz, cout, L)
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 2e-5
    up = torch.randn(bz, cout, L, generator=g)
    out.backward(up.to(DEV))
    ref.backward(up)
    assert max_err(x_d.grad.cpu().numpy(), x_c.grad.numpy()) <= 5e-5
    for c, w, b in zip(conv.list_of_conv1d, ws, bs):
        assert max_err(c.weight.grad.cpu().numpy(), w.grad.numpy()) <= 2e-4
        assert max_err(c.bias.grad.cpu().numpy(), b.grad.numpy()) <= 2e-4
    with pytest.raises(AssertionError):
        quiet(MyConv1d, [2], cin, cout)               # even width: the reference's assert (layers.py:39)


def test_lastfeat_and_fm_forward_match_oracle():
    from oracle import ref_cpu as O
    from review_based_recommender_amd.models.deepconn.layers import FM, LastFeat
    g = torch.Generator().manual_seed(4)
    bz, H, K, U, I = 9, 30, 8, 21, 17
    lf = LastFeat(U, H, K, padding_idx=0).to(DEV)
    feat = torch.randn(bz, H, generator=g)
    ids = torch.randint(0, U, (bz,), generator=g)
    ids[0] = 0
    f_d = _leaf(feat.to(DEV))
    out = lf(f_d, ids.to(DEV))
    W, b, E = _leaf(lf.W.cpu()), _leaf(lf.b.cpu()), _leaf(lf.ebd.weight.cpu())
    f_c = _leaf(feat)
    ref = O.last_feat(f_c, ids, W, b, E)
    assert out.shape == ref.shape == (bz, K)
    assert max_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    up = torch.randn(bz, K, generator=g)
    out.backward(up.to(DEV))
    ref.backward(up)
    assert max_err(f_d.grad.cpu().numpy(), f_c.grad.numpy()) <= 1e-5
    assert max_err(lf.W.grad.cpu().numpy(), W.grad.numpy()) <= 1e-5
    assert max_err(lf.b.grad.cpu().numpy(), b.grad.numpy()) <= 1e-5
    assert max_err(lf.ebd.weight.grad.cpu().numpy(), E.grad.numpy()) <= 1e-5        # pad row: zero on both sides
    assert float(lf.ebd.weight.grad[0].abs().max()) == 0.0

    fm = FM(U, I, K, 0.0, user_padding_idx=0, item_padding_idx=0).to(DEV).eval()
    uf, itf = torch.randn(bz, K, generator=g), torch.randn(bz, K, generator=g)
    uid, iid = torch.randint(0, U, (bz,), generator=g), torch.randint(0, I, (bz,), generator=g)
    u_d, i_d = _leaf(uf.to(DEV)), _leaf(itf.to(DEV))
    pred = fm(u_d, i_d, uid.to(DEV), iid.to(DEV))
    h, gb = _leaf(fm.h.cpu()), _leaf(fm.g_bias.cpu())
    ub, ib = _leaf(fm.user_bias.weight.cpu()), _leaf(fm.item_bias.weight.cpu())
    u_c, i_c = _leaf(uf), _leaf(itf)
    ref = O.fm(u_c, i_c, uid, iid, h, gb, ub, ib)
    assert pred.shape == ref.shape == (bz, 1)                 # the reference returns [bz, 1]; DeepCoNNpp views it to [bz]
    assert max_err(pred.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    pred.sum().backward()
    ref.sum().backward()
    assert max_err(u_d.grad.cpu().numpy(), u_c.grad.numpy()) <= 1e-5
    assert max_err(fm.h.grad.cpu().numpy(), h.grad.numpy()) <= 1e-5
    assert max_err(fm.user_bias.weight.grad.cpu().numpy(), ub.grad.numpy()) <= 1e-5


def _deepconn(cfg):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    m = quiet(DeepCoNNpp, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.0)
    m.load_state_dict(synth.deepconn_params(cfg, 0))
    return m.to(DEV)


@pytest.mark.parametrize("which", ["token", "user", "item", "negative"])
def test_out_of_range_id_raises_and_writes_nothing_out_of_bounds(which, conv_mode):
    """An id == table size (or < 0) must behave like nn.Embedding: IndexError (here: at the next check point), and -- the
    part a kernel has to get right -- no access outside the tables: the step with the bad id equals the step with that id
    replaced by the padding row, gradients included."""
    from review_based_recommender_amd import functional as RF
    cfg = synth.DEEPCONN_CFGS["small"]
    b = synth.deepconn_batch(cfg, 1)
    keys = ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")
    bad = {k: b[k].clone() for k in keys}
    fixed = {k: b[k].clone() for k in keys}
    if which == "token":
        bad["u_docs"][1, 3], fixed["u_docs"][1, 3] = cfg["V"], 0
        bad["u_masks"][1, 3] = fixed["u_masks"][1, 3] = True
    elif which == "negative":
        bad["i_docs"][2, 0], fixed["i_docs"][2, 0] = -7, 0
        bad["i_masks"][2, 0] = fixed["i_masks"][2, 0] = True
    elif which == "user":
        bad["u_ids"][0], fixed["u_ids"][0] = cfg["U"] + 5, 0
    else:
        bad["i_ids"][3], fixed["i_ids"][3] = cfg["I"], 0
    RF.check_id_errors()                                   # clean slate
    grads = []
    for batch in (bad, fixed):
        m = _deepconn(cfg).train()
        pred = m(*[batch[k].to(DEV) for k in keys])
        torch.nn.functional.mse_loss(pred, b["ratings"].to(DEV)).backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
        if batch is bad:
            with pytest.raises(IndexError, match="out of range"):
                RF.check_id_errors()
        RF.check_id_errors()                               # cleared / still clean
    for k in grads[0]:       # equal up to the summation order of the atomically accumulated gradients (dense mode's table scatter)
        assert torch.allclose(grads[0][k], grads[1][k], rtol=1e-5, atol=1e-7), k


def test_validate_ids_can_be_switched_off():
    from review_based_recommender_amd import functional as RF
    cfg = synth.DEEPCONN_CFGS["tiny"]
    b = synth.deepconn_batch(cfg, 1)
    keys = ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")
    m = _deepconn(cfg).eval()
    args = [b[k].to(DEV) for k in keys]
    with torch.no_grad():
        a = m(*args)
        m.validate_ids = False
        c = m(*args)
    assert torch.equal(a, c)
    RF.check_id_errors()


def test_out_of_range_ids_in_the_other_models():
    """NARRE (six id tensors, the attention tables keyed by the counterpart's ids), D-ATT and SimpleSiamese go through the
    same device-side check: a bad id anywhere raises at the next check point and the forward stays finite."""
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
    from review_based_recommender_amd.models.narre.narre import NARRE
    from review_based_recommender_amd.models.simple_siamese.simple_siamese import SimpleSiamese
    RF.check_id_errors()
    c = synth.NARRE_CFGS["small"]
    m = quiet(NARRE, c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"], c["R"], c["T"], 0.0, 0, 0, 0, None, "CNN")
    m.load_state_dict(synth.narre_params(c, 0))
    m.to(DEV).eval()
    b = synth.narre_batch(c, 1)
    keys = ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid")
    for field, bad in (("reuid", c["I"]), ("reiid", c["U"] + 3), ("i_text", c["V"]), ("u_id", -1)):
        args = {k: b[k].clone() for k in keys}
        args[field].view(-1)[1] = bad
        with torch.no_grad():
            pred = m(*[args[k].to(DEV) for k in keys])[0]
        assert torch.isfinite(pred).all()
        with pytest.raises(IndexError):
            RF.check_id_errors()
    c = synth.DATT_CFGS["small"]
    d = quiet(DualAtt, c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], 0.0, None)
    d.load_state_dict(synth.datt_params(c, 0))
    d.to(DEV).eval()
    b = synth.datt_batch(c, 1)
    u = b["u_docs"].clone(); u[0, 0] = c["V"] + 10
    with torch.no_grad():
        assert torch.isfinite(d(u.to(DEV), b["i_docs"].to(DEV))).all()
    with pytest.raises(IndexError):
        RF.check_id_errors()
    c = synth.SIAMESE_CFGS["small"]
    s = quiet(SimpleSiamese, c["D"], c["K"], c["V"], c["U"], c["I"], None, False, 0.0, 0.0, 0.0, c["UB"], c["LT"])
    s.load_state_dict(synth.siamese_params(c, 0))
    s.to(DEV).eval()
    b = synth.siamese_batch(c, 1)
    keys = ("u_revs", "i_revs", "u_word_masks", "i_word_masks", "u_rev_masks", "i_rev_masks", "u_ids", "i_ids")
    args = {k: b[k].clone() for k in keys}
    args["i_ids"][0] = c["I"]
    with torch.no_grad():
        assert torch.isfinite(s(*[args[k].to(DEV) for k in keys])[0]).all()
    with pytest.raises(IndexError):
        RF.check_id_errors()
    RF.check_id_errors()
