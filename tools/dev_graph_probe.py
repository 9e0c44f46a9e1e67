"""Development aid: record growing slices of the DeepCoNN step into a hipGraph, replay, compare with eager.

    python tools/dev_graph_probe.py            # runs the stages one by one in child processes, stops at the first failure
    python tools/dev_graph_probe.py STAGE MODE # one stage in this process
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

STAGES = ["fwd", "fwd_bwd", "fwd_bwd_clip", "full"]


def run_stage(stage, mode, cfgname="small"):
    import contextlib
    import io

    import torch
    import torch.nn.functional as F

    import synth
    from review_based_recommender_amd import _lib
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step

    _lib.lib().rbr_set_conv_mode({"dense": 1, "product": 2}[mode])
    cfg = synth.DEEPCONN_CFGS[cfgname]
    dev = torch.device("cuda", 0)

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            m = DeepCoNNpp(cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.0)
        m.load_state_dict(synth.deepconn_params(cfg, 0))
        return m.to(dev)

    def batch(seed):
        b = synth.deepconn_batch(cfg, seed)
        return tuple(b[k].to(dev) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")), b["ratings"].to(dev)

    model = build()
    a0, r0 = batch(5)
    a1, r1 = batch(6)
    static = tuple(t.clone() for t in a0)
    rs = r0.clone()

    if stage == "fwd":
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(*static)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = model(*static)
            for a in (a0, a1):
                for d, s in zip(static, a):
                    d.copy_(s)
                g.replay()
                torch.cuda.synchronize()
                ref = model(*a)
                print(stage, mode, "max |graph - eager| =", float((out - ref).abs().max()), flush=True)
        return
    model.train()
    if stage in ("fwd_bwd", "fwd_bwd_clip"):
        params = list(model.parameters())

        def body():
            for p in params:
                p.grad = None
            pred = model(*static)
            loss = F.mse_loss(pred, rs)
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(params, 5.0) if stage == "fwd_bwd_clip" else loss.detach()
            return loss.detach(), gn

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, gn = body()
        grads = [p.grad for p in params]
        for a, r in ((a0, r0), (a1, r1)):
            for d, s in zip(static, a):
                d.copy_(s)
            rs.copy_(r)
            g.replay()
            torch.cuda.synchronize()
            got = [x.clone() for x in grads]
            m2 = build()
            m2.train()
            l2 = F.mse_loss(m2(*a), r)
            l2.backward()
            if stage == "fwd_bwd_clip":
                torch.nn.utils.clip_grad_norm_(m2.parameters(), 5.0)
            err = max(float((x - p.grad).abs().max()) for x, p in zip(got, m2.parameters()))
            print(stage, mode, "loss", float(loss), float(l2), "max grad err", err, flush=True)
        return
    opt = make_optimizer(model, capturable=True)
    st = GraphedTrainStep(model, opt, a0, r0)
    m2 = build()
    m2.train()
    o2 = make_optimizer(m2)
    for a, r in ((a0, r0), (a1, r1), (a0, r0)):
        lg, gg, _ = st(a, r)
        torch.cuda.synchronize()
        le, ge, _ = train_step(m2, o2, a, r)
        err = max(float((p - q).abs().max()) for p, q in zip(model.parameters(), m2.parameters()))
        print(stage, mode, "loss", float(lg), float(le), "gnorm", float(gg), float(ge), "max param diff", err, flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        run_stage(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "small")
        sys.exit(0)
    for mode in ("dense", "product"):
        for stage in STAGES:
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), stage, mode])
            if rc != 0:
                print("STOP: stage", stage, mode, "exit", rc, flush=True)
                sys.exit(1)
    print("all stages ok", flush=True)
