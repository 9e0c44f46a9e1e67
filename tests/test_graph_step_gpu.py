"""GraphedTrainStep (hipGraph replay of the train step) against the eager train_step on the same batches."""
import pytest
import torch

import synth
from helpers import quiet

pytestmark = pytest.mark.gpu


def build_deepconn(cfg, dev, dropout=0.0):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    m = quiet(DeepCoNNpp, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, dropout)
    m.load_state_dict(synth.deepconn_params(cfg, 0))
    return m.to(dev)


def _batches(cfg, seeds, dev):
    out = []
    for s in seeds:
        b = synth.deepconn_batch(cfg, s)
        out.append((tuple(b[k].to(dev) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")),
                    b["ratings"].to(dev)))
    return out


@pytest.mark.parametrize("name", ["small", "cfg1"])
def test_graph_replay_matches_eager_steps(name, conv_mode):
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step
    cfg = synth.DEEPCONN_CFGS[name]
    dev = torch.device("cuda", 0)
    batches = _batches(cfg, [11, 12, 13], dev)
    m_e = build_deepconn(cfg, dev, dropout=0.0)
    m_g = build_deepconn(cfg, dev, dropout=0.0)
    m_e.train(); m_g.train()
    o_e = make_optimizer(m_e)
    o_g = make_optimizer(m_g, capturable=True)
    # recorded on a DIFFERENT batch than the ones replayed: token lists / work lists / argmax are device data
    cap_args, cap_r = _batches(cfg, [99], dev)[0]
    stepper = GraphedTrainStep(m_g, o_g, cap_args, cap_r)
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):      # capture + warm-up left the parameters untouched
        assert torch.equal(pe, pg)
    for args, r in batches:
        le, ge, pe_ = train_step(m_e, o_e, args, r)
        lg, gg, pg_ = stepper(args, r)
        torch.cuda.synchronize()
        assert abs(float(le) - float(lg)) <= 1e-5 * max(1.0, abs(float(le)))
        assert abs(float(ge) - float(gg)) <= 1e-4 * max(1.0, abs(float(ge)))
        assert torch.allclose(pe_, pg_, atol=1e-5, rtol=1e-5)
    for (n, pe), pg in zip(m_e.named_parameters(), m_g.parameters()):
        d = (pe - pg).abs()
        # Adam turns rounding-level gradient differences (atomic order) into +-lr steps on near-zero gradients
        assert float(d.max()) <= 1e-3 and float(d.pow(2).mean().sqrt()) <= 1e-4, n
