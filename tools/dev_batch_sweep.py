"""Dev: the cfg2 train step (hipGraph replay, HipClipAdam) at larger batches per GPU -- the per-step fixed cost (clip + Adam
over the 61 MB of parameters, the distinct-token GEMM's share that does not grow with the batch) amortises.

    python tools/dev_batch_sweep.py [256 1024 4096 ...]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
import bench
from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]
for B in sizes:
    cfg = dict(synth.DEEPCONN_CFGS["cfg2"], B=B)
    model = bench.build_model(cfg, dev)
    args, ratings = bench.batch_on(cfg, 1, dev)
    opt = make_optimizer(model, capturable=True, hip_clip_adam=True)
    step = GraphedTrainStep(model, opt, args, ratings)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    distinct = int(torch.unique(torch.cat([args[0], args[1]])).numel())
    print(json.dumps({"B": B, "ms_per_step": round(dt * 1e3, 4), "pairs_per_s": round(B / dt, 1), "distinct_tokens": distinct,
                      "positions": 2 * B * cfg["L"], "hbm_GiB_allocated": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
    del step, opt, model
    torch.cuda.empty_cache()
