"""Dev: run only the cfg2 TextCNN forward (pack + conv + finalize) N times; for rocprofv3 PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from oracle import ref_cpu as O   # only for conv_params() key parsing in this dev tool
import review_based_recommender_amd.functional as RF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS[cfgname]
p = synth.deepconn_params(cfg, 0); b = synth.deepconn_batch(cfg, 1)
ws, bs = O.conv_params(p)
tb = p["word_embeddings.embedding.weight"].to(dev)
ids = torch.cat([b["u_docs"], b["i_docs"]]).to(dev); mask = torch.cat([b["u_masks"], b["i_masks"]]).to(dev)
w_ = [w.to(dev) for w in ws]; b_ = [x.to(dev) for x in bs]
for _ in range(3): RF.textcnn(tb, ids, mask, w_, b_)
torch.cuda.synchronize()
RF.TIMER.start()
for _ in range(n): RF.textcnn(tb, ids, mask, w_, b_)
torch.cuda.synchronize()
print(RF.TIMER.summary())
