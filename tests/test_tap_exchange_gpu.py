"""Data-parallel tap exchange of the table gradient: two ranks on one GPU (gloo), see tests/dp_tap_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("exchange", ["taps", "owner"])
def test_tap_exchange_matches_dense_allreduce(exchange):
    """taps: every rank rebuilds the whole averaged table gradient from the gathered taps.  owner: rank r rebuilds the rows of
    the tokens t % N == r, the slabs are all-gathered and HipClipAdam reads them in place (row form)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", RBR_TEST_EXCHANGE=exchange)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if exchange == "taps" else "29537", os.path.join(HERE, "dp_tap_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "TAP EXCHANGE OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_rebuild_from_taps_matches_float64_reference_and_is_order_free():
    """rbr_textcnn_dtable_from_taps at the cfg2 shape with three ranks' worth of taps: every path of the rebuild is taken --
    wave-per-token rows (<= 16 taps), workgroup rows, and the hottest tokens split over several workgroups that meet in a
    global fixed-point row (> 4096 taps) -- and compared with G @ Wprod^T accumulated in float64 from the same taps.
    Two runs must agree bit for bit (the sums do not depend on the order the atomics land in)."""
    import ctypes as C

    import torch

    import synth
    from review_based_recommender_amd import _lib
    L_ = _lib.lib()
    dev = torch.device("cuda:0")
    cfg = synth.DEEPCONN_CFGS["cfg2"]
    p = synth.deepconn_params(cfg, 0)
    kz, ch = [3, 5, 7], [50, 50, 50]
    ws = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.weight"].to(dev) for i in range(3)]
    V, D = p["word_embeddings.embedding.weight"].shape
    n_docs, Ccount, KF = 2 * cfg["B"], sum(ch), max(kz)
    d = _lib.make_desc(n_docs, cfg["L"], D, V, kz, ch, 0, 0, 0)
    n = L_.rbr_textcnn_taps_count(C.byref(d))
    assert n == n_docs * Ccount * KF
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    n_sets = 3
    tok = torch.empty(n_sets * n, dtype=torch.int32, device=dev)
    val = torch.empty(n_sets * n, dtype=torch.float32, device=dev)
    for s in range(n_sets):
        b = synth.deepconn_batch(cfg, 200 + s)
        ids = torch.cat([b["u_docs"], b["i_docs"]]).to(dev)
        mask = torch.cat([b["u_masks"], b["i_masks"]]).to(dev).view(torch.uint8)
        feat = torch.rand(n_docs, Ccount, generator=g).to(dev)
        dfeat = (torch.randn(n_docs, Ccount, generator=g) * 1e-2).to(dev)
        lens = mask.sum(1, keepdim=True).clamp(min=1)
        argmax = (torch.rand(n_docs, Ccount, generator=g).to(dev) * lens).to(torch.int32)
        _lib.check(L_.rbr_textcnn_bwd_taps(C.byref(d), ids.data_ptr(), mask.data_ptr(), feat.data_ptr(), argmax.data_ptr(),
                                           dfeat.data_ptr(), tok[s * n:].data_ptr(), val[s * n:].data_ptr(), st), "taps")
    counts = torch.bincount(tok[tok >= 0].long(), minlength=V)
    assert int(counts.max()) > 3 * 4096 and int(((counts > 0) & (counts <= 16)).sum()) > 1000 and int((counts > 16).sum()) > 1000
    wsb = torch.empty(L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(d), n_sets), dtype=torch.uint8, device=dev)
    W = _lib.ptr_array(ws, torch.float32, "w")
    outs = []
    for _ in range(2):
        dtable = torch.full((V, D), float("nan"), device=dev)
        _lib.check(L_.rbr_textcnn_dtable_from_taps(C.byref(d), n_sets, tok.data_ptr(), val.data_ptr(), W, wsb.data_ptr(),
                                                   dtable.data_ptr(), st), "rebuild")
        outs.append(dtable)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    # reference: column of tap e = ((doc * C + c) * KF + j) is poff[w] + j * ch[w] + (c - ch_off[w]); WT[col, :] = W_w[cl, :, j]
    e = torch.arange(n_sets * n, device=dev) % n
    j, c = e % KF, (e // KF) % Ccount
    w = c // 50
    col = torch.tensor([0, 150, 400], device=dev)[w] + j * 50 + (c - 50 * w)
    keep = tok >= 0
    KG = sum(k * h for k, h in zip(kz, ch))
    G = torch.zeros(V * KG, dtype=torch.float64, device=dev)
    G.index_add_(0, tok[keep].long() * KG + col[keep], val[keep].double())
    WT = torch.cat([wi.permute(2, 0, 1).reshape(-1, D) for wi in ws]).double()          # [(w, j, cl), D]
    ref = (G.view(V, KG) @ WT) / n_sets
    got = outs[0].double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    assert float(got[counts == 0].abs().max()) == 0.0
    # the owner partition of the same rebuild: rank r's slab holds the rows of the tokens t % n_sets == r, the same bits as the
    # replicated rebuild (fixed-point sums for the hot rows; the cold rows are summed in array order, and the stable compaction
    # + stable sort keep a token's taps in the order the replicated sort leaves them in)
    v_own = L_.rbr_textcnn_taps_owner_rows(C.byref(d), n_sets)
    assert v_own == (V + n_sets - 1) // n_sets
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    merged = torch.full((v_own * n_sets, D), float("nan"), device=dev)
    for r in range(n_sets):
        slab = torch.full((v_own, D), float("nan"), device=dev)
        _lib.check(L_.rbr_textcnn_dtable_from_taps_owner(C.byref(d), n_sets, r, tok.data_ptr(), val.data_ptr(), W, wsb.data_ptr(),
                                                         slab.data_ptr(), flag.data_ptr(), st), "owner rebuild")
        merged[r::n_sets] = slab
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert torch.equal(merged[:V], outs[0])
    assert float(merged[V:].abs().max()) == 0.0 if merged.shape[0] > V else True
    # a rank that owns more taps than the sort is sized for reports it instead of building a wrong slab silently
    tok_skew = torch.where(tok >= 0, (tok // n_sets) * n_sets, torch.full_like(tok, n_sets))      # every entry a tap on a token of rank 0
    slab = torch.empty(v_own, D, device=dev)
    _lib.check(L_.rbr_textcnn_dtable_from_taps_owner(C.byref(d), n_sets, 0, tok_skew.data_ptr(), val.data_ptr(), W, wsb.data_ptr(),
                                                     slab.data_ptr(), flag.data_ptr(), st), "owner rebuild (skewed)")
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
