"""Dev: cost of rebuilding the averaged table gradient from n_sets ranks' taps on one GPU, cfg2 shape, taps taken from n_sets
different synthetic batches: the replicated rebuild (rbr_textcnn_dtable_from_taps) and the compute legs of the owner-partitioned
one (rbr_textcnn_dtable_from_taps_owner for the busiest rank, the slab's sum of squares, and the optimizer's row-form pass over
the gathered slabs against its dense pass).  python tools/dev_taps_bench.py [n_sets ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from review_based_recommender_amd import _lib
L_ = _lib.lib()
dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS["cfg2"]
p = synth.deepconn_params(cfg, 0)
ws = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.weight"].to(dev) for i in range(3)]
V, D = p["word_embeddings.embedding.weight"].shape
d = _lib.make_desc(2 * cfg["B"], cfg["L"], D, V, [3, 5, 7], [50, 50, 50], 0, 0, 0)
n = L_.rbr_textcnn_taps_count(C.byref(d))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
for n_sets in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    tok = torch.empty(n_sets * n, dtype=torch.int32, device=dev); val = torch.empty(n_sets * n, dtype=torch.float32, device=dev)
    for s in range(n_sets):
        b = synth.deepconn_batch(cfg, 100 + s)
        ids = torch.cat([b["u_docs"], b["i_docs"]]).to(dev); mask = torch.cat([b["u_masks"], b["i_masks"]]).to(dev).view(torch.uint8)
        feat = torch.rand(ids.shape[0], 150, generator=g).to(dev); dfeat = (torch.randn(ids.shape[0], 150, generator=g) * 1e-2).to(dev)
        lens = mask.sum(1, keepdim=True).clamp(min=1)
        argmax = (torch.rand(ids.shape[0], 150, generator=g).to(dev) * lens).to(torch.int32)
        assert L_.rbr_textcnn_bwd_taps(C.byref(d), ids.data_ptr(), mask.data_ptr(), feat.data_ptr(), argmax.data_ptr(), dfeat.data_ptr(),
                                       tok[s * n:].data_ptr(), val[s * n:].data_ptr(), st) == 0
    wsb = torch.empty(L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(d), n_sets), dtype=torch.uint8, device=dev)
    dtable = torch.empty(V, D, device=dev)
    W = _lib.ptr_array(ws, torch.float32, "w")
    for _ in range(3):
        assert L_.rbr_textcnn_dtable_from_taps(C.byref(d), n_sets, tok.data_ptr(), val.data_ptr(), W, wsb.data_ptr(), dtable.data_ptr(), st) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    host = []
    evs[0].record()
    for k in range(20):
        h0 = time.perf_counter()
        L_.rbr_textcnn_dtable_from_taps(C.byref(d), n_sets, tok.data_ptr(), val.data_ptr(), W, wsb.data_ptr(), dtable.data_ptr(), st)
        host.append(time.perf_counter() - h0)
        evs[k + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(20))
    print(f"n_sets={n_sets}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per rebuild (events: median {per[10]:.1f}, max {per[-1]:.1f}; "
          f"host call median {sorted(host)[10] * 1e6:.0f}, max {max(host) * 1e6:.0f} us); union tokens {int((tok >= 0).sum())} taps, "
          f"{int(torch.unique(tok[tok >= 0]).numel())} distinct; payload {n * 8 / 1e6:.1f} MB per rank", flush=True)

    # ---- owner partition: every rank's share of the taps, and the rebuild of the busiest one
    own_cnt = [int(((tok >= 0) & (tok % n_sets == r)).sum()) for r in range(n_sets)]
    worst = max(range(n_sets), key=lambda r: own_cnt[r])
    v_own = L_.rbr_textcnn_taps_owner_rows(C.byref(d), n_sets)
    slab = torch.empty(v_own, D, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def owner_once():
        assert L_.rbr_textcnn_dtable_from_taps_owner(C.byref(d), n_sets, worst, tok.data_ptr(), val.data_ptr(), W, wsb.data_ptr(),
                                                     slab.data_ptr(), flag.data_ptr(), st) == 0
    for _ in range(3):
        owner_once()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(20):
        owner_once()
    e1.record()
    for _ in range(20):
        sq = torch.linalg.vector_norm(slab).square()
    e2.record()
    torch.cuda.synchronize()
    cap = min(n_sets * n, 2 * ((n_sets * n + n_sets - 1) // n_sets) + 4096)
    print(f"  owner: shares of the {int((tok >= 0).sum())} taps by rank {own_cnt} (sort sized for {cap}; overflow flag {int(flag.item())}); "
          f"rebuild of rank {worst}'s {v_own} rows {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, slab norm {e1.elapsed_time(e2) / 20 * 1e3:.1f} us; "
          f"slab all-gather payload {(v_own + 1) * D * 4 / 1e6:.2f} MB per rank", flush=True)
