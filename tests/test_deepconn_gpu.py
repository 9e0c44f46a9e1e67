"""DeepCoNN++ on the HIP path vs golden vectors captured from the reference and vs the CPU oracle.
Every test goes through the C ABI (ctypes -> librbr_hip.so)."""
import numpy as np
import pytest
import torch

import synth
from helpers import check_grads, check_params_after, golden, max_err, quiet

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4   # BASELINE.json north_star: outputs within 1e-4 (fp32) of the reference CPU forward


def _model(cfg, sd, arch="CNN", dropout=0.0):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    kz = cfg["kz"] if arch == "CNN" else cfg["kz"][:1]
    m = quiet(DeepCoNNpp, cfg["U"], cfg["I"], cfg["V"], kz, cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, dropout, arch)
    m.load_state_dict(sd)
    return m.to("cuda:0")


def _batch(b):
    d = torch.device("cuda:0")
    return tuple(b[k].to(d) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")), b["ratings"].to(d)


@pytest.mark.parametrize("name,cfgname,edge", [
    ("deepconn_tiny", "tiny", True), ("deepconn_small", "small", True), ("deepconn_k3", "k3", False),
    ("deepconn_cfg1", "cfg1", False), ("deepconn_cfg2", "cfg2", False)])
def test_deepconn_matches_reference(golden_dir, name, cfgname, edge, conv_mode):
    from review_based_recommender_amd.train_step import make_optimizer, train_step
    g = golden(golden_dir, name)
    cfg = synth.DEEPCONN_CFGS[cfgname]
    model = _model(cfg, synth.deepconn_params(cfg, 0))
    args, ratings = _batch(synth.deepconn_batch(cfg, 1, edge_cases=edge))

    model.eval()
    with torch.no_grad():
        pred = model(*args)
    assert pred.shape == (cfg["B"],) and pred.dtype == torch.float32
    assert max_err(pred.cpu().numpy(), g["pred_eval"]) <= FWD_TOL

    model.train()
    opt = make_optimizer(model)
    for step in range(3):
        loss, gnorm, pred = train_step(model, opt, args, ratings)
        if step == 0:
            assert max_err(pred.cpu().numpy(), g["pred"]) <= FWD_TOL
            assert abs(float(loss) - float(g["loss"])) <= 1e-4          # |MSE_hip - MSE_ref| <= 1e-4
            assert abs(float(gnorm) - float(g["gnorm"])) <= 2e-4 * float(g["gnorm"])
        if step in (0, 2):
            check_params_after(model, g, f"after{step + 1}")
    assert abs(float(loss) - float(g["loss_after3"])) <= 2e-4 * max(1.0, float(g["loss_after3"]))


@pytest.mark.parametrize("name,cfgname,edge", [
    ("deepconn_tiny", "tiny", True), ("deepconn_small", "small", True), ("deepconn_k3", "k3", False),
    ("deepconn_cfg1", "cfg1", False), ("deepconn_cfg2", "cfg2", False)])
def test_deepconn_gradients_match_reference(golden_dir, name, cfgname, edge, conv_mode):
    g = golden(golden_dir, name)
    cfg = synth.DEEPCONN_CFGS[cfgname]
    model = _model(cfg, synth.deepconn_params(cfg, 0))
    args, ratings = _batch(synth.deepconn_batch(cfg, 1, edge_cases=edge))
    model.train()
    loss = torch.nn.functional.mse_loss(model(*args), ratings)
    loss.backward()
    check_grads({k: p.grad for k, p in model.named_parameters()}, g)
    # nn.Embedding(padding_idx=0): the pad row never receives gradient
    assert float(model.word_embeddings.embedding.weight.grad[0].abs().max()) == 0.0
    assert float(model.user_feat.ebd.weight.grad[0].abs().max()) == 0.0


def test_ngram_features_match_oracle(golden_dir, conv_mode):
    """TextCNN features alone (u_rev_feats / i_rev_feats of the reference forward)."""
    from review_based_recommender_amd import functional as RF
    from oracle import ref_cpu as O
    for name, cfgname in (("deepconn_small", "small"), ("deepconn_cfg1", "cfg1")):
        g = golden(golden_dir, name)
        cfg = synth.DEEPCONN_CFGS[cfgname]
        p = synth.deepconn_params(cfg, 0)
        b = synth.deepconn_batch(cfg, 1, edge_cases=(cfgname == "small"))
        ws, bs = O.conv_params(p)
        d = torch.device("cuda:0")
        feat, am = RF.textcnn(p["word_embeddings.embedding.weight"].to(d), b["u_docs"].to(d), b["u_masks"].to(d),
                              [w.to(d) for w in ws], [x.to(d) for x in bs], return_argmax=True)
        assert max_err(feat.cpu().numpy(), g["u_rev_feats"]) <= 2e-5
        assert int(am.min()) >= 0 and int(am.max()) < cfg["L"]


def test_state_dict_keys_and_shapes_match_reference():
    cfg = synth.DEEPCONN_CFGS["tiny"]
    sd = synth.deepconn_params(cfg, 0)
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    m = quiet(DeepCoNNpp, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.5)
    ours = m.state_dict()
    assert list(ours.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(ours[k].shape) == tuple(sd[k].shape), k


def test_dropout_train_mode_runs_and_eval_is_deterministic():
    cfg = synth.DEEPCONN_CFGS["small"]
    model = _model(cfg, synth.deepconn_params(cfg, 0), dropout=0.5)
    args, _ = _batch(synth.deepconn_batch(cfg, 1))
    model.train()
    a = model(*args)
    b = model(*args)
    assert not torch.equal(a, b)           # FM dropout (layers.py:202) draws a fresh mask
    model.eval()
    assert torch.equal(model(*args), model(*args))


def test_error_conventions():
    """AssertionError for even widths / H not divisible (layers.py:38-39), ValueError for unknown arch (:116)."""
    from review_based_recommender_amd.models.deepconn.layers import NgramFeat
    with pytest.raises(AssertionError):
        quiet(NgramFeat, [2], 8, 6, 16)
    with pytest.raises(AssertionError):
        quiet(NgramFeat, [3, 5], 8, 7, 16)
    with pytest.raises(ValueError):
        quiet(NgramFeat, [3], 8, 6, 16, arch="nope")
    m = quiet(NgramFeat, "3,5", 8, 6, 16)    # "3,5" strings are accepted (layers.py:34-36)
    assert m.feature_layer[0].kernel_sizes == [3, 5]


def test_cpu_tensors_are_rejected():
    """There is no CPU fallback: the product path refuses host tensors."""
    from review_based_recommender_amd import functional as RF
    cfg = synth.DEEPCONN_CFGS["tiny"]
    p = synth.deepconn_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1)
    with pytest.raises(RuntimeError):
        RF.textcnn(p["word_embeddings.embedding.weight"], b["u_docs"], b["u_masks"],
                   [p["ngram.feature_layer.0.list_of_conv1d.0.weight"]], [p["ngram.feature_layer.0.list_of_conv1d.0.bias"]])


def _dup_batch(cfg):
    """A batch that hits few users / items, each always with the same document (as the doc split guarantees)."""
    b = synth.deepconn_batch(cfg, 1)
    u_ids = torch.tensor([1, 2, 1, 3, 2, 1, 3, 3]); i_ids = torch.tensor([4, 4, 5, 5, 4, 6, 6, 4])
    b["u_ids"], b["i_ids"] = u_ids, i_ids
    b["u_docs"], b["u_masks"] = b["u_docs"][u_ids], b["u_masks"][u_ids]
    b["i_docs"], b["i_masks"] = b["i_docs"][i_ids], b["i_masks"][i_ids]
    return b


def test_dedup_by_id_matches_the_oracle_on_the_duplicated_batch(conv_mode):
    """f-3: with dedup_by_id each distinct user / item document is encoded once; predictions, loss and EVERY gradient must
    equal the CPU oracle run on the full duplicated batch (the reference re-encodes per pair, deepconn.py:46-47)."""
    from oracle import ref_cpu as O
    cfg = synth.DEEPCONN_CFGS["small"]
    b = _dup_batch(cfg)
    keys = ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")
    sd = synth.deepconn_params(cfg, 0)
    hist = O.train_steps(sd, lambda q: O.deepconn_forward(q, *[b[k] for k in keys]), b["ratings"], n_steps=1)[0]
    args, ratings = _batch(b)
    model = _model(cfg, sd)
    model.dedup_by_id = True
    model.train()
    pred = model(*args)
    loss = torch.nn.functional.mse_loss(pred, ratings)
    loss.backward()
    assert max_err(pred.detach().cpu().numpy(), hist["pred"].numpy()) <= 1e-5
    assert abs(float(loss) - float(hist["loss"])) <= 1e-5
    for k, p in model.named_parameters():
        ref = hist["grads"][k]
        scale = float(ref.norm()) + 1e-12
        assert float((p.grad.cpu() - ref).abs().max()) <= 2e-6 + 2e-4 * scale, k


@pytest.fixture
def product_mode():
    from review_based_recommender_amd import _lib
    _lib.lib().rbr_set_conv_mode(2)          # the token-product formulation (and with it the fused step) also at the small shape
    yield
    _lib.lib().rbr_set_conv_mode(0)


def test_dedup_step_is_graph_capturable_and_skips_repeated_documents(product_mode):
    """The first-occurrence pass is a device kernel with static shapes (no torch.unique, no host sync): the dedup step
    replays as a hipGraph on new batches, and the blanked repeated rows are the ones that are not first occurrences."""
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step
    cfg = synth.DEEPCONN_CFGS["small"]
    b = _dup_batch(cfg)
    args, ratings = _batch(b)
    first, masks = RF.dedup_rows(args[4], args[5], cfg["U"], cfg["I"], torch.cat([args[2], args[3]]), cfg["L"])
    B = cfg["B"]
    assert first.tolist() == [0, 1, 0, 3, 1, 0, 3, 3] + [B + x for x in [0, 0, 2, 2, 0, 5, 5, 0]]
    keep = torch.tensor([f == r for r, f in enumerate(first.tolist())], device=masks.device)
    assert not masks[~keep].any() and torch.equal(masks[keep], torch.cat([args[2], args[3]])[keep])

    sd = synth.deepconn_params(cfg, 0)
    m_g, m_e = _model(cfg, sd), _model(cfg, sd)
    m_g.dedup_by_id = m_e.dedup_by_id = True
    m_g.train(); m_e.train()
    o_g, o_e = make_optimizer(m_g, hip_clip_adam=True), make_optimizer(m_e, hip_clip_adam=True)
    stepper = GraphedTrainStep(m_g, o_g, args, ratings, keep_graph=True)
    n_launch = stepper.kernel_launches()
    # the dedup rides the fused step now -- dedup_mark + dedup_apply + the gradient fold on top of the 14-15
    # launches of the fused DeepCoNN step (it used to fall back to the un-fused ~25-launch step with a torch index_select)
    assert n_launch is None or n_launch <= 19, n_launch
    for step in range(3):
        b2 = _dup_batch(cfg)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(step))      # another duplicate pattern every step
        for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids", "ratings"):
            b2[k] = b2[k][perm]
        a2, r2 = _batch(b2)
        lg, _, _ = stepper(a2, r2)
        le, _, _ = train_step(m_e, o_e, a2, r2)
        torch.cuda.synchronize()
        assert abs(float(lg) - float(le)) <= 1e-5 * max(1.0, abs(float(le))), (step, float(lg), float(le))


def test_dataparallel_wrapper_does_not_crash():
    """the trainers may wrap the model in nn.DataParallel (train_deepconn_pp.py:129-131); on one visible GPU
    the wrapper calls the module directly -- forward and backward must still work and parameters keep their names."""
    cfg = synth.DEEPCONN_CFGS["small"]
    model = _model(cfg, synth.deepconn_params(cfg, 0))
    dp = torch.nn.DataParallel(model, device_ids=[0])
    args, ratings = _batch(synth.deepconn_batch(cfg, 1))
    pred = dp(*args)
    assert pred.shape == (cfg["B"],)
    torch.nn.functional.mse_loss(pred, ratings).backward()
    assert all(p.grad is not None for p in model.parameters())
    assert [k[len("module."):] for k in dp.state_dict().keys()] == list(model.state_dict().keys())
