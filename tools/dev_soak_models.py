"""Dev: replay the recorded train step of a secondary model many times on a fixed batch: the loss must stay finite and fall,
the step time must not drift.      python tools/dev_soak_models.py datt|narre [N]"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer

dev = torch.device("cuda:0")
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
with contextlib.redirect_stdout(io.StringIO()):
    if which == "datt":
        from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
        c = synth.DATT_CFGS["cfg4"]
        m = DualAtt(c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], 0.5, None)
        m.load_state_dict(synth.datt_params(c, 0, table_scale=0.3))
        b = synth.datt_batch(c, 1)
        args = (b["u_docs"].to(dev), b["i_docs"].to(dev))
    else:
        from review_based_recommender_amd.models.narre.narre import NARRE
        c = synth.NARRE_CFGS["cfg3"]
        m = NARRE(c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"], c["R"], c["T"], 0.5, 0, 0, 0, None, "CNN")
        m.load_state_dict(synth.narre_params(c, 0))
        b = synth.narre_batch(c, 1)
        args = tuple(b[k].to(dev) for k in ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid"))
m.to(dev).train()
st = GraphedTrainStep(m, make_optimizer(m, hip_clip_adam=True), args, b["ratings"].to(dev))
losses, times = [], []
for blk in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 6):
        out = st()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) / (n // 6) * 1e3)
    losses.append(float(out[0]))
print(which, "loss per block", [round(x, 4) for x in losses])
print(which, "ms/step per block", [round(x, 4) for x in times])
assert all(x == x and x < 1e6 for x in losses) and losses[-1] < losses[0]
assert max(times) < 1.15 * min(times)
for p in m.parameters():
    assert torch.isfinite(p).all()
print("soak ok")
