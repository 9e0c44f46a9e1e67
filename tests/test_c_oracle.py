"""The scalar C restatement (oracle/textcnn_ref.c) against the torch oracle (oracle/ref_cpu.py):
forward values / argmax and -- the point of the exercise -- that the max-pool-SPARSE backward the HIP
kernels implement equals autograd's dense backward of the reference's op sequence.  CPU only."""
import numpy as np
import torch
import torch.nn.functional as F

import synth
from oracle import c_ref
from oracle import ref_cpu as O


def _np(ts):
    return [np.ascontiguousarray(t.detach().numpy()) for t in ts]


def test_c_forward_and_sparse_backward_match_autograd():
    for cfgname, edge in (("tiny", True), ("small", True)):
        cfg = synth.DEEPCONN_CFGS[cfgname]
        p = synth.deepconn_params(cfg, 0)
        b = synth.deepconn_batch(cfg, 1, edge_cases=edge)
        table = p["word_embeddings.embedding.weight"].clone().requires_grad_(True)
        ws, bs = O.conv_params(p)
        ws = [w.clone().requires_grad_(True) for w in ws]
        bs = [x.clone().requires_grad_(True) for x in bs]
        ids, mask = b["i_docs"], b["i_masks"]
        feat = O.ngram_feat_cnn(O.word_embedding(table, ids), mask, ws, bs)
        gen = torch.Generator().manual_seed(3)
        d_feat = torch.randn(feat.shape, generator=gen)
        feat.backward(d_feat)

        tn, wn, bn = _np([table])[0], _np(ws), _np(bs)
        f_c, am_c = c_ref.textcnn_fwd(ids.numpy(), mask.numpy(), None, tn, wn, bn)
        np.testing.assert_allclose(f_c, feat.detach().numpy(), atol=2e-6)
        dWs, dbs, dtable, _ = c_ref.textcnn_bwd_sparse(ids.numpy(), mask.numpy(), None, tn, wn, f_c, am_c,
                                                       np.ascontiguousarray(d_feat.numpy()))
        for a, t in zip(dWs, ws):
            np.testing.assert_allclose(a, t.grad.numpy(), atol=5e-6)
        for a, t in zip(dbs, bs):
            np.testing.assert_allclose(a, t.grad.numpy(), atol=5e-6)
        np.testing.assert_allclose(dtable, table.grad.numpy(), atol=5e-6)
        assert np.abs(dtable[0]).max() == 0.0      # padding_idx row


def test_c_valid_tanh_gated_variant_matches_autograd():
    """The D-ATT flavour: per-token gate, 'valid' convs k=2,3,4, tanh, pool over L-k+1, no mask."""
    rng = np.random.default_rng(5)
    n_docs, L, E, V, C = 3, 14, 6, 12, 4
    ids = torch.from_numpy(rng.integers(0, V, size=(n_docs, L)).astype(np.int64))
    table = torch.from_numpy(rng.standard_normal((V, E)).astype(np.float32)).requires_grad_(True)
    gate = torch.from_numpy(rng.uniform(0.1, 0.9, size=(n_docs, L)).astype(np.float32)).requires_grad_(True)
    ws = [torch.from_numpy(rng.standard_normal((C, E, k)).astype(np.float32) * 0.3).requires_grad_(True) for k in (2, 3, 4)]
    bs = [torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.1).requires_grad_(True) for _ in range(3)]
    x = (F.embedding(ids, table) * gate.unsqueeze(-1)).permute(0, 2, 1)
    outs = [F.max_pool1d(torch.tanh(F.conv1d(x, w, b_)), L - w.shape[2] + 1).squeeze(-1) for w, b_ in zip(ws, bs)]
    feat = torch.cat(outs, dim=1)
    d_feat = torch.from_numpy(rng.standard_normal(feat.shape).astype(np.float32))
    feat.backward(d_feat)
    tn, gn = _np([table, gate])
    f_c, am_c = c_ref.textcnn_fwd(ids.numpy(), None, gn, tn, _np(ws), _np(bs), pad_valid=True, act=1)
    np.testing.assert_allclose(f_c, feat.detach().numpy(), atol=2e-6)
    dWs, dbs, dtable, dgate = c_ref.textcnn_bwd_sparse(ids.numpy(), None, gn, tn, _np(ws), f_c, am_c,
                                                       np.ascontiguousarray(d_feat.numpy()), pad_valid=True, act=1,
                                                       padding_idx=-1)
    for a, t in zip(dWs, ws):
        np.testing.assert_allclose(a, t.grad.numpy(), atol=5e-6)
    np.testing.assert_allclose(dtable, table.grad.numpy(), atol=5e-6)
    np.testing.assert_allclose(dgate, gate.grad.numpy(), atol=5e-6)
