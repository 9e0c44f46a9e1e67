"""Round-3 step fusions against the paths they replace and against the reference fixtures:
  * compact row gradient of the word table (RBR_G_ROWS -> rbr_clip_adam_step_rows) vs the dense gradient;
  * the fused encoder + head (+ MSE) function vs the separate textcnn / pair_head / mse_loss functions;
  * the configuration bench.py times -- GraphedTrainStep + HipClipAdam -- at cfg2 / cfg3 / cfg4 vs the reference's golden step
    (trainer/train_deepconn_pp.py:161-168 is what the fixtures recorded)."""
import os

import numpy as np
import pytest
import torch

import synth
from helpers import check_grads, check_params_after, golden, max_err, quiet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")


def _deepconn(cfg, dropout=0.0):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    m = quiet(DeepCoNNpp, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, dropout)
    m.load_state_dict(synth.deepconn_params(cfg, 0))
    return m.to(DEV)


def _batch(cfg, seed, edge=False):
    b = synth.deepconn_batch(cfg, seed, edge_cases=edge)
    return tuple(b[k].to(DEV) for k in KEYS), b["ratings"].to(DEV)


class _env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("cfgname", ["small", "cfg1", "cfg2"])
def test_row_gradient_equals_the_dense_gradient_bit_for_bit(cfgname):
    """HipClipAdam(row_grads=True) takes the table gradient as the rows of the batch's tokens.  Fed the SAME gradient in dense
    form (the rows scattered into a zero [V, D] tensor), a second optimizer on a copy of the model ends its steps with the
    same bits in parameters and Adam state while the clip coefficient is 1; the norm itself is summed in another order
    (per-workgroup partials of the producer against chunks of the dense tensor), so it -- and with it a clipping step --
    agrees to rounding, not to the bit.
    (Two separate backwards cannot be compared bit for bit: G is built with f32 atomics.)"""
    from review_based_recommender_amd import _lib
    from review_based_recommender_amd.train_step import HipClipAdam, _forward_loss_backward
    _lib.lib().rbr_set_conv_mode(2)
    try:
        cfg = synth.DEEPCONN_CFGS[cfgname]
        ma, mb = _deepconn(cfg), _deepconn(cfg)
        ma.train(); mb.train()
        table_a = ma.word_embeddings.embedding.weight
        keep, HipClipAdam.ROW_GRAD_MIN_ROWS = HipClipAdam.ROW_GRAD_MIN_ROWS, 1      # the small fixtures' tables qualify too
        try:
            oa = HipClipAdam(list(ma.parameters()), lr=2e-3, row_grads=True)
        finally:
            HipClipAdam.ROW_GRAD_MIN_ROWS = keep
        ob = HipClipAdam(list(mb.parameters()), lr=2e-3, row_grads=False)
        for step in range(3):
            args, r = _batch(cfg, 3 + step, edge=(cfgname == "small"))
            oa.zero_grad()
            _forward_loss_backward(ma, args, r, oa)
            assert table_a.grad is None and table_a in oa._row_grads, "the compact row gradient was not handed over"
            dense = oa._row_grads[table_a].to_dense()
            # what the dense backward computes: the same rows, zeros elsewhere (checked against a second, dense backward)
            ob.zero_grad()
            _forward_loss_backward(mb, args, r)
            ref = mb.word_embeddings.embedding.weight.grad
            assert float((dense - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1e-30)
            assert torch.equal(dense == 0, ref == 0) or float(((dense == 0) != (ref == 0)).float().mean()) < 1e-3
            for pa, pb in zip(ma.parameters(), mb.parameters()):
                pb.grad = dense.clone() if pa is table_a else pa.grad.clone()
            clip = 0.05 if step == 2 else 1e9           # steps 0, 1: coefficient exactly 1; step 2: clipped
            ga = oa.clip_and_step(clip).clone()
            gb = ob.clip_and_step(clip).clone()
            assert abs(float(ga) - float(gb)) <= 1e-6 * float(gb), (step, float(ga), float(gb))
            oa.materialize_grads()
            for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
                if step < 2:
                    assert torch.equal(pa, pb), (step, k)
                    assert torch.equal(pa.grad, pb.grad), (step, k, "gradient left in place")
                    assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]), (step, k)
                    assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), (step, k)
                else:
                    assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7), (step, k)      # an ulp of the clip coefficient
                    assert torch.allclose(pa.grad, pb.grad, rtol=1e-5, atol=0), (step, k, "clipped gradient")
                    assert torch.allclose(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"], rtol=1e-5, atol=1e-12), (step, k)
                    assert torch.allclose(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"], rtol=1e-5, atol=1e-20), (step, k)
                    with torch.no_grad():
                        pb.copy_(pa)                     # (the test ends here; kept for symmetry)
        oa.close()
    finally:
        _lib.lib().rbr_set_conv_mode(0)


def test_narre_row_gradient_through_the_row_gemm_equals_the_sparse_product():
    """NARRE cfg3 (many short documents: G is a third full): the compact row gradient is G @ Wprod^T on the bf16 pipe with exact
    three-plane splits (csrc/textcnn_prod_b16.hip: prod_b16_rows_gemm), its sums of squares come out of the same launch.  Against the
    dense gradient of a plain backward, which still takes the sparse row product: rows to rounding, zeros elsewhere, norm to 1e-5."""
    from review_based_recommender_amd.models.narre.narre import NARRE
    from review_based_recommender_amd.train_step import HipClipAdam, _forward_loss_backward
    cfg = synth.NARRE_CFGS["cfg3"]
    keys = ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid")
    b = synth.narre_batch(cfg, 9)
    args, ratings = tuple(b[k].to(DEV) for k in keys), b["ratings"].to(DEV)
    ms = []
    for _ in range(2):
        m = quiet(NARRE, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["H"], cfg["D"], cfg["A"], cfg["K"], cfg["R"], cfg["T"], 0.0,
                  0, 0, 0, None, "CNN")
        m.load_state_dict(synth.narre_params(cfg, 0))
        ms.append(m.to(DEV).train())
    ma, mb = ms
    ta, tb = ma.word_embeddings.embedding.weight, mb.word_embeddings.embedding.weight
    oa = HipClipAdam(list(ma.parameters()), lr=2e-3)
    oa.zero_grad()
    _forward_loss_backward(ma, args, ratings, oa)
    assert ta.grad is None and ta in oa._row_grads, "NARRE's table gradient is expected in row form"
    rg = oa._row_grads[ta]
    dense, sq = rg.to_dense(), float(rg.sq.double().sum())
    _forward_loss_backward(mb, args, ratings)
    torch.cuda.synchronize()
    ref = tb.grad
    scale = float(ref.abs().max())
    assert float((dense - ref).abs().max()) <= 2e-6 * scale + 1e-12, (float((dense - ref).abs().max()), scale)
    assert abs(sq - float((ref.double() ** 2).sum())) <= 1e-5 * sq
    for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):          # the other gradients are the same kernels: same to rounding
        if pa is not ta:
            assert float((pa.grad - pb.grad).abs().max()) <= 1e-6 + 2e-5 * float(pb.grad.abs().max()), k


def test_hipclipadam_then_torch_adam_on_one_model_updates_the_table():
    """a HipClipAdam that exists beside another optimizer on the same model must not swallow the word-table
    gradient.  Two graph-replayed HipClipAdam steps, then eager clip_grad_norm_ + torch.optim.Adam steps on the SAME model
    (tools/bench_models.py's order), against a fresh model that takes the same batches through HipClipAdam (eager) and torch Adam:
    after every torch step the table has moved, with the gradient of THAT batch (not one batch stale)."""
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step
    cfg = synth.DEEPCONN_CFGS["cfg1"]
    keep_rows = None
    from review_based_recommender_amd.train_step import HipClipAdam
    keep_rows, HipClipAdam.ROW_GRAD_MIN_ROWS = HipClipAdam.ROW_GRAD_MIN_ROWS, 1
    try:
        ma, mb = _deepconn(cfg), _deepconn(cfg)
        ma.train(); mb.train()
        ta, tb = ma.word_embeddings.embedding.weight, mb.word_embeddings.embedding.weight
        batches = [_batch(cfg, 30 + k) for k in range(5)]
        ga = make_optimizer(ma, hip_clip_adam=True)
        stepper = GraphedTrainStep(ma, ga, *batches[0])
        gb = make_optimizer(mb, hip_clip_adam=True)
        for k in range(2):
            stepper(*batches[k])
            train_step(mb, gb, *batches[k])
        torch.cuda.synchronize()
        assert float((ta - tb).abs().max()) <= 2.5e-3          # (+-lr steps on rounding-level gradients, as everywhere)
        # ga stays alive, as in tools/bench_models.py; the eager steps belong to torch's Adam
        oa, ob = make_optimizer(ma), make_optimizer(mb)
        for k in range(2, 5):
            before = ta.detach().clone()
            train_step(ma, oa, *batches[k])
            assert ta.grad is not None, "the backward of a torch-Adam step must leave a dense table gradient"
            train_step(mb, ob, *batches[k])
            torch.cuda.synchronize()
            assert float((ta.grad - tb.grad).abs().max()) <= 2e-6 + 2e-3 * float(tb.grad.abs().max()), k
            moved_a, moved_b = (ta.detach() - before).abs().sum(1) > 0, tb.grad.abs().sum(1) > 0
            # the rows of THIS batch moved (a fresh Adam moves an element by ~lr wherever its gradient is not vanishingly small)
            assert int((moved_a & moved_b).sum()) >= 0.98 * int(moved_b.sum()) > 0, (k, int(moved_a.sum()), int(moved_b.sum()))
            if k == 2:          # a fresh Adam: nothing but this batch's rows can move (later steps also carry momentum of earlier rows)
                assert not bool((moved_a & ~moved_b).any()), (int(moved_a.sum()), int(moved_b.sum()))
        del stepper
    finally:
        HipClipAdam.ROW_GRAD_MIN_ROWS = keep_rows


@pytest.mark.parametrize("cfgname,edge", [("small", True), ("cfg1", False), ("cfg2", False)])
def test_fused_encoder_head_equals_the_separate_functions(cfgname, edge):
    """DeepCoNNpp.forward through functional.encode_head (id check in the prepare launch, pool epilogue + head + MSE in one
    launch, G cleared by the gather launch) vs the separate sanitize_ids / textcnn / pair_head / mse_loss functions
    (RBR_FUSED_STEP=0): same predictions and loss bit for bit, same gradients (the table's up to f32 atomic order)."""
    from review_based_recommender_amd import _lib, functional as RF
    from review_based_recommender_amd.train_step import _forward_loss_backward
    _lib.lib().rbr_set_conv_mode(2)
    try:
        cfg = synth.DEEPCONN_CFGS[cfgname]
        ma, mb = _deepconn(cfg), _deepconn(cfg)
        ma.train(); mb.train()
        args, r = _batch(cfg, 1, edge=edge)
        assert ma._fused_ok(args[0], args[1])
        pa, la = _forward_loss_backward(ma, args, r)
        with _env(RBR_FUSED_STEP="0"):
            assert not mb._fused_ok(args[0], args[1])
            pb, lb = _forward_loss_backward(mb, args, r)
        torch.cuda.synchronize()
        assert torch.equal(pa, pb)
        assert float(la) == float(lb)
        for (k, qa), qb in zip(ma.named_parameters(), mb.parameters()):
            if k == "word_embeddings.embedding.weight":
                scale = float(qb.grad.abs().max()) + 1e-30
                assert float((qa.grad - qb.grad).abs().max()) <= 1e-5 * scale, k
            else:
                assert torch.allclose(qa.grad, qb.grad, rtol=1e-5, atol=1e-7 * (float(qb.grad.abs().max()) + 1e-30)), k
        # eval forward (no grad): fused and separate agree bit for bit as well
        ma.eval(); mb.eval()
        with torch.no_grad():
            ea = ma(*args)
            with _env(RBR_FUSED_STEP="0"):
                eb = mb(*args)
        assert torch.equal(ea, eb)
        RF.check_id_errors()
    finally:
        _lib.lib().rbr_set_conv_mode(0)


def test_fused_path_reports_out_of_range_ids():
    """The id range check that rides in the fused prepare launch raises the same IndexError at the next check point."""
    from review_based_recommender_amd import _lib, functional as RF
    cfg = synth.DEEPCONN_CFGS["cfg1"]
    m = _deepconn(cfg).eval()
    args, _ = _batch(cfg, 1)
    _lib.lib().rbr_set_conv_mode(2)
    try:
        assert m._fused_ok(args[0], args[1])
        bad = list(args)
        bad[0] = bad[0].clone()
        bad[0][0, 0] = cfg["V"]                    # one past the table
        with torch.no_grad():
            m(*bad)
        with pytest.raises(IndexError):
            RF.check_id_errors()
        with torch.no_grad():
            m(*args)
        RF.check_id_errors()                       # clean again
    finally:
        _lib.lib().rbr_set_conv_mode(0)


def _graphed_vs_golden(model, g, args, ratings, cap_args, cap_ratings, tol_max=1e-3, slots=1):
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer
    model.train()
    opt = make_optimizer(model, hip_clip_adam=True)
    stepper = GraphedTrainStep(model, opt, cap_args, cap_ratings, slots=slots)      # recorded on a DIFFERENT batch
    for k in range(slots):
        stepper.stage(k, args, ratings)
    for step in range(3):
        # slots > 1: the fixture's batch sits in every input slot and the slots' graphs take turns (no per-step copy)
        loss, gnorm, pred = stepper(args, ratings) if slots == 1 else stepper(slot=step % slots)
        torch.cuda.synchronize()
        if step == 0:
            assert max_err(pred.cpu().numpy(), g["pred"]) <= 1e-4
            assert abs(float(loss) - float(g["loss"])) <= 1e-4
            assert abs(float(gnorm) - float(g["gnorm"])) <= 2e-4 * float(g["gnorm"])
        if step in (0, 2):
            check_params_after(model, g, f"after{step + 1}", tol_max=tol_max)
    return opt


@pytest.mark.parametrize("slots", [1, 2])
def test_graphed_hipclipadam_cfg2_matches_reference(golden_dir, slots):
    """The chain bench.py times -- GraphedTrainStep(model, HipClipAdam) replaying the fused step, one recorded graph per input
    slot -- on the cfg2 fixture: loss, clipped norm, predictions and the parameters after 1 and 3 steps against the
    reference's recording."""
    g = golden(golden_dir, "deepconn_cfg2")
    cfg = synth.DEEPCONN_CFGS["cfg2"]
    model = _deepconn(cfg)
    args, ratings = _batch(cfg, 1)
    cap_args, cap_r = _batch(cfg, 77)
    opt = _graphed_vs_golden(model, g, args, ratings, cap_args, cap_r, slots=slots)
    table = model.word_embeddings.embedding.weight
    assert table in opt._row_grads, "the benched step is expected to hand the table gradient over in row form"


def test_graphed_hipclipadam_narre_cfg3_matches_reference(golden_dir):
    from review_based_recommender_amd.models.narre.narre import NARRE
    g = golden(golden_dir, "narre_cfg3")
    cfg = synth.NARRE_CFGS["cfg3"]
    m = quiet(NARRE, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["H"], cfg["D"], cfg["A"], cfg["K"], cfg["R"], cfg["T"], 0.0,
              0, 0, 0, None, "CNN")
    m.load_state_dict(synth.narre_params(cfg, 0))
    m = m.to(DEV)
    keys = ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid")
    b, c = synth.narre_batch(cfg, 1), synth.narre_batch(cfg, 77)
    args, ratings = tuple(b[k].to(DEV) for k in keys), b["ratings"].to(DEV)
    cap_args, cap_r = tuple(c[k].to(DEV) for k in keys), c["ratings"].to(DEV)
    _graphed_vs_golden(m, g, args, ratings, cap_args, cap_r)


def test_graphed_hipclipadam_datt_cfg4_matches_reference(golden_dir):
    from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
    g = golden(golden_dir, "datt_cfg4")
    cfg = synth.DATT_CFGS["cfg4"]
    m = quiet(DualAtt, cfg["V"], cfg["L"], cfg["win"], cfg["l_out"], cfg["g_out"], cfg["E"], cfg["h1"], cfg["h2"], 0.0, None)
    m.load_state_dict(synth.datt_params(cfg, 0, table_scale=0.3))       # as the fixture (tests/test_narre_datt_gpu.py)
    m = m.to(DEV)
    b, c = synth.datt_batch(cfg, 1), synth.datt_batch(cfg, 77)
    args, ratings = (b["u_docs"].to(DEV), b["i_docs"].to(DEV)), b["ratings"].to(DEV)
    cap_args, cap_r = (c["u_docs"].to(DEV), c["i_docs"].to(DEV)), c["ratings"].to(DEV)
    _graphed_vs_golden(m, g, args, ratings, cap_args, cap_r)


@pytest.mark.parametrize("conv", ["dense", "product"])
def test_fixed_dtable_mode_is_bit_reproducible(conv):
    """functional.set_dtable_mode("fixed") (env RBR_DTABLE_MODE=fixed): the word-table gradient of two backward runs over the
    same batch is the same bits (64-bit fixed-point cell sums: SURVEY 7's deterministic mode; the reference's CPU embedding
    backward, models/deepconn/layers.py:22-24, is reproducible too), and agrees with the default atomics path to rounding."""
    from review_based_recommender_amd import _lib, functional as RF
    _lib.lib().rbr_set_conv_mode({"dense": 1, "product": 2}[conv])
    try:
        cfg = synth.DEEPCONN_CFGS["cfg1"]
        args, r = _batch(cfg, 5)
        grads = []
        for mode in ("fixed", "fixed", None):
            RF.set_dtable_mode(mode)
            with _env(RBR_FUSED_STEP="0"):
                m = _deepconn(cfg)
                m.train()
                torch.nn.functional.mse_loss(m(*args), r).backward()
            grads.append(m.word_embeddings.embedding.weight.grad.clone())
        assert torch.equal(grads[0], grads[1])
        assert float(grads[0][0].abs().max()) == 0.0                 # padding row
        scale = float(grads[2].abs().max())
        assert float((grads[0] - grads[2]).abs().max()) <= 1e-5 * scale
    finally:
        RF.set_dtable_mode(None)
        _lib.lib().rbr_set_conv_mode(0)
