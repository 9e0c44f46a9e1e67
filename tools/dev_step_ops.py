"""Dev: list every device kernel of one eager cfg2 train step in launch order, with the aten / autograd op that launched
it and its input shapes (torch.profiler) -- to find the small launches worth folding.

    python tools/dev_step_ops.py [deepconn|narre]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
import bench
from review_based_recommender_amd.train_step import make_optimizer, train_step
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS["cfg2"]
model = bench.build_model(cfg, dev)
args, ratings = bench.batch_on(cfg, 1, dev)
opt = make_optimizer(model, hip_clip_adam=True)
for _ in range(3):
    train_step(model, opt, args, ratings)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(model, opt, args, ratings)
    torch.cuda.synchronize()
ops = sorted([e for e in prof.events() if getattr(e, "kernels", None)], key=lambda e: e.time_range.start)
total = 0.0
for op in ops:
    for k in op.kernels:
        total += k.duration
        print(f"{k.duration:8.1f} us  {k.name[:72]:72s} <- {op.name[:40]} {str(op.input_shapes)[:90]}")
print(f"{sum(len(o.kernels) for o in ops)} kernels, {total:.1f} us")
