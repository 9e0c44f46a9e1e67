"""Dev: kernel launches of one recorded train step.  Run under `rocprofv3 --kernel-trace --stats --output-format csv` with two
replay counts and divide the difference of the total call counts by the difference of the counts:
    python tools/dev_count_launches.py datt|narre|deepconn N"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer

dev = torch.device("cuda:0")
which, n = sys.argv[1], int(sys.argv[2])
with contextlib.redirect_stdout(io.StringIO()):
    if which == "datt":
        from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
        c = synth.DATT_CFGS["cfg4"]
        m = DualAtt(c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], 0.5, None)
        m.load_state_dict(synth.datt_params(c, 0, table_scale=0.3))
        b = synth.datt_batch(c, 1)
        args = (b["u_docs"].to(dev), b["i_docs"].to(dev))
    elif which == "narre":
        from review_based_recommender_amd.models.narre.narre import NARRE
        c = synth.NARRE_CFGS["cfg3"]
        m = NARRE(c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"], c["R"], c["T"], 0.5, 0, 0, 0, None, "CNN")
        m.load_state_dict(synth.narre_params(c, 0))
        b = synth.narre_batch(c, 1)
        args = tuple(b[k].to(dev) for k in ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid"))
    else:
        from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
        c = synth.DEEPCONN_CFGS["cfg2"]
        m = DeepCoNNpp(c["U"], c["I"], c["V"], c["kz"], c["D"], c["H"], c["K"], c["L"], None, 0.5)
        m.load_state_dict(synth.deepconn_params(c, 0))
        b = synth.deepconn_batch(c, 1)
        args = tuple(b[k].to(dev) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids"))
m.to(dev).train()
st = GraphedTrainStep(m, make_optimizer(m, hip_clip_adam=True), args, b["ratings"].to(dev))
for _ in range(n):
    st()
torch.cuda.synchronize()
print("replayed", n)
