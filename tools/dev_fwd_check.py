import os, sys, time
os.environ["RBR_DEV_PARTIAL_LIB"]="1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import torch, numpy as np
import synth
from oracle import ref_cpu as O
import review_based_recommender_amd.functional as RF
dev = torch.device("cuda:0")
for name, edge in (("tiny", True), ("small", True), ("k3", False), ("cfg1", False), ("cfg2", False)):
    cfg = synth.DEEPCONN_CFGS[name]
    p = synth.deepconn_params(cfg, 0); b = synth.deepconn_batch(cfg, 1, edge_cases=edge)
    ws, bs = O.conv_params(p)
    table = p["word_embeddings.embedding.weight"]
    ids = torch.cat([b["u_docs"], b["i_docs"]]); mask = torch.cat([b["u_masks"], b["i_masks"]])
    with torch.no_grad():
        ref = O.ngram_feat_cnn(O.word_embedding(table, ids), mask, ws, bs)
    t0=time.time()
    feat, am = RF.textcnn(table.to(dev), ids.to(dev), mask.to(dev), [w.to(dev) for w in ws], [x.to(dev) for x in bs], return_argmax=True)
    torch.cuda.synchronize()
    err = (feat.cpu() - ref).abs().max().item()
    print(name, "max err", err, "ref max", ref.abs().max().item(), "t", time.time()-t0, flush=True)
# timing cfg2
tb=table.to(dev); i_=ids.to(dev); m_=mask.to(dev); w_=[w.to(dev) for w in ws]; b_=[x.to(dev) for x in bs]
for _ in range(3): RF.textcnn(tb,i_,m_,w_,b_)
torch.cuda.synchronize()
RF.TIMER.start()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): RF.textcnn(tb,i_,m_,w_,b_)
e1.record(); torch.cuda.synchronize()
print("cfg2 textcnn fwd total ms/iter", e0.elapsed_time(e1)/10, RF.TIMER.summary())
ms = RF.TIMER.summary()["textcnn_conv_fwd"][1]
print("conv TFLOP/s", 117.9648e9/ (ms*1e-3)/1e12)
