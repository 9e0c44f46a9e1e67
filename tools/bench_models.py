"""Secondary measurements (not the driver's bench line): NARRE cfg3, D-ATT cfg4 and SimpleSiamese (defalut_simple_train.json
shape, batch 256) forward / train step on one GPU, plus the CPU oracle's forward on the same batch.
python tools/bench_models.py [narre|datt|siamese|all] [--cpu] [--no-graph] [--precision=bf16]"""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, synth
from review_based_recommender_amd import _lib
from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step

dev = torch.device("cuda:0")
PRECISION = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--precision=")), None)   # f32 | bf16x3 | bf16x2 | bf16
if PRECISION:
    from review_based_recommender_amd import functional as _RF
    _RF.set_prod_precision(PRECISION)


def quiet(fn, *a):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a)


def run(name, model, args, ratings, B, flops_fwd, bytes_fwd_per_pair):
    model.train()
    graph_ms = None
    if "--no-graph" not in sys.argv:      # the step replayed as a hipGraph (how bench.py times DeepCoNN)
        gopt = make_optimizer(model, hip_clip_adam=True)
        stepper = GraphedTrainStep(model, gopt, args, ratings)
        for _ in range(5):
            stepper()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            stepper()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / 30 * 1e3
    opt = make_optimizer(model)
    for _ in range(5):
        train_step(model, opt, args, ratings)
    torch.cuda.synchronize()
    _lib.TIMER.start()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        train_step(model, opt, args, ratings)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / n
    _lib.TIMER.stop()
    ks = _lib.TIMER.summary()
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(*args)
        torch.cuda.synchronize()
        fwd_eager = fwd = (time.perf_counter() - t0) / n
        fwd_graph = None
        if "--no-graph" not in sys.argv:      # the same forward replayed from a hipGraph: GPU time without the host's launch cost
            from review_based_recommender_amd.train_step import GraphedForward
            gf = GraphedForward(model, args)
            for _ in range(3):
                gf()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                gf()
            torch.cuda.synchronize()
            fwd_graph = fwd = (time.perf_counter() - t0) / n
    print(json.dumps({"model": name, "conv_precision": PRECISION or "bf16x3 (default)", "train_graph_ms": None if graph_ms is None else round(graph_ms, 3),
                      "train_graph_pairs_per_s": None if graph_ms is None else round(B / graph_ms * 1e3, 1),
                      "train_ms": round(step * 1e3, 3), "train_pairs_per_s": round(B / step, 1),
                      "fwd_ms": round(fwd * 1e3, 3), "fwd_pairs_per_s": round(B / fwd, 1),
                      "fwd_eager_ms": round(fwd_eager * 1e3, 3), "fwd_launch": "hipGraph replay" if fwd_graph is not None else "eager",
                      "fwd_TFLOPs_algorithmic": round(flops_fwd / fwd / 1e12, 2),
                      # algorithmic bytes of one forward (ids + masks + gathered rows) against the HBM peak
                      "roofline": {"bound": "hbm", "achieved": round(bytes_fwd_per_pair * B / fwd / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(bytes_fwd_per_pair * B / fwd / 8e12, 4),
                                   "bytes_per_pair": bytes_fwd_per_pair, "what": "eval forward, whole model (no single dominant kernel)"},
                      "kernels_ms": {k: round(v[1], 4) for k, v in ks.items()}}))


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("narre", "all"):
    from review_based_recommender_amd.models.narre.narre import NARRE
    c = synth.NARRE_CFGS["cfg3"]
    m = quiet(NARRE, c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"], c["R"], c["T"], 0.5, 0, 0, 0, None, "CNN")
    m.load_state_dict(synth.narre_params(c, 0)); m.to(dev)
    b = synth.narre_batch(c, 1)
    args = tuple(b[k].to(dev) for k in ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid"))
    run("NARRE cfg3 (B=256, 10x50 tok/side, D=300, fp32)", m, args, b["ratings"].to(dev), c["B"], 270.0e6 * c["B"], 2 * c["R"] * c["T"] * (8 + 1 + 4 * c["D"]) + 2 * c["R"] * 8)
    if "--cpu" in sys.argv:
        from oracle import ref_cpu as O
        p = synth.narre_params(c, 0)
        a = tuple(b[k] for k in ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid"))
        with torch.no_grad():
            O.narre_forward(p, *a); t0 = time.perf_counter(); O.narre_forward(p, *a); print("cpu fwd pairs/s", c["B"] / (time.perf_counter() - t0), "threads", torch.get_num_threads())
if which in ("datt", "all"):
    from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
    c = synth.DATT_CFGS["cfg4"]
    m = quiet(DualAtt, c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], 0.5, None)
    m.load_state_dict(synth.datt_params(c, 0, table_scale=0.3)); m.to(dev)
    b = synth.datt_batch(c, 1)
    args = (b["u_docs"].to(dev), b["i_docs"].to(dev))
    run("D-ATT cfg4 (B=512, 2x1024 tok, E=100, fp32)", m, args, b["ratings"].to(dev), c["B"], 453.3e6 * c["B"], 2 * c["L"] * (8 + 4 * c["E"]))
    if "--cpu" in sys.argv:
        from oracle import ref_cpu as O
        p = synth.datt_params(c, 0, table_scale=0.3)
        with torch.no_grad():
            O.datt_forward(p, b["u_docs"], b["i_docs"]); t0 = time.perf_counter(); O.datt_forward(p, b["u_docs"], b["i_docs"]); print("cpu fwd pairs/s", c["B"] / (time.perf_counter() - t0), "threads", torch.get_num_threads())

if which in ("siamese", "all"):
    from review_based_recommender_amd.models.simple_siamese.simple_siamese import SimpleSiamese
    c = dict(synth.SIAMESE_CFGS["toys"], B=256)
    m = quiet(SimpleSiamese, c["D"], c["K"], c["V"], c["U"], c["I"], None, False, 0.5, 0.2, 0.0, c["UB"], c["LT"])
    m.load_state_dict(synth.siamese_params(c, 0)); m.to(dev)
    b = synth.siamese_batch(c, 1)
    keys = ("u_revs", "i_revs", "u_word_masks", "i_word_masks", "u_rev_masks", "i_rev_masks", "u_ids", "i_ids")
    args = tuple(b[k].to(dev) for k in keys)
    # algorithmic work: 2 towers x R x T row adds of D floats per pair (a gather, not a contraction)
    run("SimpleSiamese (B=256, 11x50 tok/side, D=108, fp32)", m, args, b["ratings"].to(dev), c["B"], 2.0 * c["R"] * c["T"] * c["D"] * c["B"], 2 * c["R"] * c["T"] * (8 + 1 + 4 * c["D"]))
    if "--cpu" in sys.argv:
        from oracle import ref_cpu as O
        p = synth.siamese_params(c, 0)
        a = tuple(b[k] for k in keys)
        with torch.no_grad():
            O.siamese_forward(p, *a); t0 = time.perf_counter(); O.siamese_forward(p, *a); print("cpu fwd pairs/s", c["B"] / (time.perf_counter() - t0), "threads", torch.get_num_threads())
