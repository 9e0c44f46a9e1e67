import sys, time, os, contextlib, io
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, synth
from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
c = synth.DATT_CFGS["cfg4"]
with contextlib.redirect_stdout(io.StringIO()):
    m = DualAtt(c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], 0.5, None)
m.load_state_dict(synth.datt_params(c, 0, table_scale=0.3)); m.to("cuda:0").eval()
b = synth.datt_batch(c, 1)
args = (b["u_docs"].to("cuda:0"), b["i_docs"].to("cuda:0"))
if "--train-first" in sys.argv:
    from review_based_recommender_amd.train_step import make_optimizer, train_step, GraphedTrainStep
    m.train()
    r = b["ratings"].to("cuda:0")
    if "--graph" in sys.argv:
        st = GraphedTrainStep(m, make_optimizer(m, hip_clip_adam=True), args, r)
        for _ in range(5): st()
    opt = make_optimizer(m)
    for _ in range(5): train_step(m, opt, args, r)
    torch.cuda.synchronize()
    m.eval()
with torch.no_grad():
    for _ in range(3): m(*args)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): m(*args)
        torch.cuda.synchronize()
        print("fwd ms", (time.perf_counter() - t0) / 20 * 1e3)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(5): m(*args)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=12))
