"""Dev: 300 rebuilds of the averaged table gradient from 8 ranks' worth of taps (cfg2 shape), each compared bit for bit with the
first: the split-token path's last-arrival protocol and the integer atomics must give the same bits whatever the scheduling.
python tools/dev_taps_soak.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from review_based_recommender_amd import _lib
L_ = _lib.lib(); dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS["cfg2"]; p = synth.deepconn_params(cfg, 0)
ws = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.weight"].to(dev) for i in range(3)]
V, D = p["word_embeddings.embedding.weight"].shape
d = _lib.make_desc(2 * cfg["B"], cfg["L"], D, V, [3, 5, 7], [50, 50, 50], 0, 0, 0)
n = L_.rbr_textcnn_taps_count(C.byref(d)); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0); n_sets = 8
tok = torch.empty(n_sets * n, dtype=torch.int32, device=dev); val = torch.empty(n_sets * n, dtype=torch.float32, device=dev)
for s in range(n_sets):
    b = synth.deepconn_batch(cfg, 100 + s)
    ids = torch.cat([b["u_docs"], b["i_docs"]]).to(dev); mask = torch.cat([b["u_masks"], b["i_masks"]]).to(dev).view(torch.uint8)
    feat = torch.rand(ids.shape[0], 150, generator=g).to(dev); dfeat = (torch.randn(ids.shape[0], 150, generator=g) * 1e-2).to(dev)
    lens = mask.sum(1, keepdim=True).clamp(min=1); argmax = (torch.rand(ids.shape[0], 150, generator=g).to(dev) * lens).to(torch.int32)
    assert L_.rbr_textcnn_bwd_taps(C.byref(d), ids.data_ptr(), mask.data_ptr(), feat.data_ptr(), argmax.data_ptr(), dfeat.data_ptr(), tok[s * n:].data_ptr(), val[s * n:].data_ptr(), st) == 0
wsb = torch.empty(L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(d), n_sets), dtype=torch.uint8, device=dev)
W = _lib.ptr_array(ws, torch.float32, "w")
ref = None; bad = 0
for it in range(300):
    dt = torch.full((V, D), float("nan"), device=dev)
    assert L_.rbr_textcnn_dtable_from_taps(C.byref(d), n_sets, tok.data_ptr(), val.data_ptr(), W, wsb.data_ptr(), dt.data_ptr(), st) == 0
    if ref is None: ref = dt.clone()
    elif not torch.equal(ref, dt): bad += 1
torch.cuda.synchronize()
print("soak: 300 rebuilds at 8 sets,", bad, "differed; finite:", bool(torch.isfinite(ref).all()))
