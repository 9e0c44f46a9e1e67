"""Dev: a few eager cfg2 train steps with UNIFORM random token ids instead of the Zipf ids of bench.py -- run under
rocprofv3 --kernel-trace --stats to see which kernels owe their time to hot tokens (same-address atomics in build_g)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
import bench
from review_based_recommender_amd.train_step import make_optimizer, train_step

dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS["cfg2"]
model = bench.build_model(cfg, dev)
args, ratings = bench.batch_on(cfg, 1, dev)
args = list(args)
g = torch.Generator().manual_seed(0)
for k in (0, 1):
    args[k] = torch.randint(1, cfg["V"], args[k].shape, generator=g).to(dev)
opt = make_optimizer(model, hip_clip_adam=True)
for _ in range(12):
    train_step(model, opt, tuple(args), ratings)
torch.cuda.synchronize()
print("distinct tokens", int(torch.unique(torch.cat([args[0], args[1]])).numel()))
