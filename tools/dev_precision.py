"""Dev: the token-product GEMM in each arithmetic (rbr_set_prod_precision) -- TextCNN features against an fp64 CPU
evaluation of the same conv, and the HIP-event time of the prod_table stage.  python tools/dev_precision.py [cfg ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from oracle import ref_cpu as O   # dev tool: conv_params() key parsing and the fp64 reference
import review_based_recommender_amd.functional as RF
from review_based_recommender_amd import _lib
dev = torch.device("cuda:0")
L_ = _lib.lib()
L_.rbr_set_conv_mode(2)
for cfgname in (sys.argv[1:] or ["small", "k3", "cfg1", "cfg2"]):
    cfg = synth.DEEPCONN_CFGS[cfgname]
    p = synth.deepconn_params(cfg, 0); b = synth.deepconn_batch(cfg, 1, edge_cases=cfgname in ("tiny", "small"))
    ws, bs = O.conv_params(p)
    table = p["word_embeddings.embedding.weight"]
    ids = torch.cat([b["u_docs"], b["i_docs"]]); mask = torch.cat([b["u_masks"], b["i_masks"]])
    with torch.no_grad():
        ref64 = O.ngram_feat_cnn(O.word_embedding(table.double(), ids), mask, [w.double() for w in ws], [x.double() for x in bs])
        ref32 = O.ngram_feat_cnn(O.word_embedding(table, ids), mask, ws, bs)
    print(f"{cfgname}: torch CPU f32 vs fp64: {(ref32.double() - ref64).abs().max().item():.3e}", flush=True)
    tb = table.to(dev); i_ = ids.to(dev); m_ = mask.to(dev); w_ = [w.to(dev) for w in ws]; b_ = [x.to(dev) for x in bs]
    for name, mode in _lib.PROD_PRECISIONS.items():
        L_.rbr_set_prod_precision(mode)
        feat = RF.textcnn(tb, i_, m_, w_, b_)
        torch.cuda.synchronize()
        err = (feat.cpu().double() - ref64).abs().max().item()
        for _ in range(3): RF.textcnn(tb, i_, m_, w_, b_)
        torch.cuda.synchronize()
        RF.TIMER.start()
        for _ in range(20): RF.textcnn(tb, i_, m_, w_, b_)
        torch.cuda.synchronize()
        RF.TIMER.stop()
        s = RF.TIMER.summary()
        print(f"  {name:7s} max |feat - fp64| {err:.3e}   prepare {s['textcnn_prod_prepare'][1]*1e3:7.1f} us  table "
              f"{s['textcnn_prod_table'][1]*1e3:7.1f} us  pool {s['textcnn_prod_pool'][1]*1e3:7.1f} us", flush=True)
    L_.rbr_set_prod_precision(-1)
