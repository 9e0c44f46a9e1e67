#!/usr/bin/env python3
"""bench.py -- (user,item) pairs/s of the DeepCoNN review-encoder train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]

With --gpus N > 1 and no WORLD_SIZE in the environment, bench.py starts its own ranks BEFORE touching the GPU: a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
(the same command a launcher would use; under such a launcher bench.py simply runs as one rank), relays rank 0's JSON
line and exits with the child's code.

Workload = BASELINE.json configs[1] ("cfg2"): DeepCoNN, batch 256 pairs per GPU, 2 x 512-token
documents per pair, 300-d embeddings, conv widths 3/5/7 x 50 channels, latent 32, V = 50 002,
fp32, synthetic Zipf token ids (tests/golden/synth.py).  One step = the reference trainer's step
(trainer/train_deepconn_pp.py:161-168): zero_grad -> forward -> MSELoss -> backward ->
[N>1: RCCL gradient exchange] -> clip_grad_norm_(5.0) -> Adam(lr=2e-3).  Inputs are resident in
HBM before the timed region.  Weak scaling: every rank processes its own 256 pairs.

Timing.  The step is recorded into hipGraphs (train_step.GraphedTrainStep) and replayed.  Every step takes the next of
--batches (default 4) DISTINCT resident batches; each batch sits in its own input slot of the step (one recorded graph per
slot: a loader stages batch i + 1 into the next slot while slot i runs), so a step starts with no copy of its inputs.
`copy_per_step` is the one-slot form of earlier rounds (one device-to-device copy into the single input block inside the
timed region).  A timed block is EXACTLY --steps steps between
barrier + synchronize on both sides, max over ranks; the block is repeated until >= 0.5 s of timed region and
`ms_per_step` / `value` are the MEDIAN block (`ms_per_step_min`, `repeats` beside them).  Secondary results, same step:
`single_batch_replay` (one batch resident in the input block, no copy: round 1's number), `uniform_ids` (token ids
uniform over the vocabulary) and `all_distinct` (every vocabulary entry occurs: distinct tokens = min(V, positions)),
each with its distinct-token count -- the token-product conv's cost follows that count.

Rank 0 prints ONE JSON line.  `roofline` prices the longest kernel of the step, measured live with HIP events around
its C-ABI launches (events cannot be read inside a replayed graph: the same K steps are launched eagerly right after the
timed region for them); `roofline_gemm` prices the distinct-token GEMM of the conv (the kernel round 1 priced) on the
MFMA pipe it runs on.  `cpu_baseline` times the CPU oracle (oracle/ref_cpu.py, torch CPU ops) on the same workload on
this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

import synth  # noqa: E402

WORKLOAD = "cfg2"
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0 # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBPS = 8000.0         # same guide, "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured with a float4 copy)
MIN_TIMED_S = float(os.environ.get("RBR_BENCH_MIN_TIMED_S", "0.5"))


def conv_fwd_flops(cfg) -> float:
    """Algorithmic FLOPs of one conv-kernel launch: 2 docs x 2*L*D*C_w*sum(kz) per pair."""
    per_w = cfg["H"] // len(cfg["kz"])
    return 2.0 * cfg["B"] * 2.0 * cfg["L"] * cfg["D"] * per_w * sum(cfg["kz"])


def active_tile_fraction(cfg, masks) -> float:
    """Fraction of 32-token wave-tiles the conv kernel actually computes: a tile whose tokens (halo included)
    are all masked is skipped because its conv sums are exactly 0 (csrc/textcnn_fwd.hip, tile_scan_kernel).
    Recomputed here on the host with the same rule, for the executed-FLOP figure of the roofline object."""
    kf = max(cfg["kz"])
    p = (kf - 1) // 2
    m = masks.cpu()
    n_docs, L = m.shape
    wpd = (L + 31) // 32
    act = 0
    for w in range(wpd):
        lo, hi = max(0, w * 32 - p), min(L, w * 32 + 32 + kf - 1 - p)
        act += int(m[:, lo:hi].any(dim=1).sum())
    return act / float(n_docs * wpd)


def measured_traffic(name="r01_conv_fwd_pmc.json"):
    """HBM bytes per launch of the priced kernel from the committed rocprofv3 PMC passes (profiles/<name>:
    FETCH_SIZE x2 + WRITE_SIZE, separate passes, gfx950 correction) -- not measurable from inside this
    process; None when the file is absent."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_launch_corrected"])
    except (OSError, KeyError, ValueError):
        return None


def build_model(cfg, device):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = DeepCoNNpp(cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["D"], cfg["H"], cfg["K"], cfg["L"], None, 0.5)
    m.load_state_dict(synth.deepconn_params(cfg, 0))
    return m.to(device)


def batch_on(cfg, seed, device, ids="zipf"):
    """A resident batch.  ids: "zipf" (synth.py: the headline), "uniform" (token ids uniform over [2, V)) or
    "all_distinct" (the unmasked positions walk the whole vocabulary: distinct tokens = min(V - 2, unmasked positions))."""
    b = synth.deepconn_batch(cfg, seed)
    if ids != "zipf":
        import numpy as np
        rng = np.random.default_rng(1000 + seed)
        for k in ("u_docs", "i_docs"):
            keep = b[k] != 0
            n = int(keep.sum())
            if ids == "uniform":
                new = rng.integers(2, cfg["V"], size=n)
            else:
                new = 2 + (rng.permutation(n) + (0 if k == "u_docs" else n)) % (cfg["V"] - 2)
            b[k] = b[k].clone()
            b[k][keep] = torch.from_numpy(new.astype("int64"))
    args = tuple(b[k].to(device) for k in ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids"))
    return args, b["ratings"].to(device)


def cpu_baseline(cfg, budget_s=20.0):
    """The oracle's train step (same math, torch CPU ops) on a bounded sample: whole batches of the
    same workload until ~budget_s of CPU time is spent (at least 2 timed steps after 1 warm-up)."""
    import torch.nn.functional as F
    from oracle import ref_cpu as O
    p = synth.deepconn_params(cfg, 0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=2e-3)
    b = synth.deepconn_batch(cfg, 1)
    args = (b["u_docs"], b["i_docs"], b["u_masks"], b["i_masks"], b["u_ids"], b["i_ids"])

    def step():
        opt.zero_grad()
        pred = O.deepconn_forward(leaves, *args, dropout_p=0.5, training=True)
        loss = F.mse_loss(pred, b["ratings"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(leaves.values()), 5.0)
        opt.step()

    # pick the thread count that is fastest for this op mix on this host (more threads is not always faster
    # for oneDNN convs at this size): one timed forward per candidate after a warm-up forward
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, 64, default_threads) if t <= max(ncpu, 1)})
    best_t, best_s = default_threads, float("inf")
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            O.deepconn_forward(p, *args)
            f0 = time.perf_counter()
            O.deepconn_forward(p, *args)
            dt = time.perf_counter() - f0
            if dt < best_s:
                best_t, best_s = t, dt
    torch.set_num_threads(best_t)
    step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and (el >= budget_s or n >= 50):
            break
    with torch.no_grad():
        f0 = time.perf_counter()
        O.deepconn_forward(p, *args)
        fwd_s = time.perf_counter() - f0
    torch.set_num_threads(default_threads)
    return {
        "value": round(cfg["B"] * n / el, 2), "unit": "pairs/s", "cores": best_t, "kind": "port",
        "sample": f"{n} full train steps of the same cfg2 batch (B=256) with oracle/ref_cpu.py, torch CPU fp32, "
                  f"{el:.1f} s wall, {best_t} threads (fastest of {cands}); host has {ncpu} logical CPUs",
        "fwd_pairs_per_s": round(cfg["B"] / fwd_s, 2),
    }


def verify_tap_exchange(model, args, ratings, grad_sync, dist, mode="auto", optimizer=None):
    """Start-up self-check of the tap exchanges (distributed.TapExchange: "taps" = every rank rebuilds the whole averaged
    gradient from the gathered taps; "owner" = rank r rebuilds the rows of its tokens and the slabs are all-gathered) on this
    job's own ranks: one backward whose gradients are all-reduced densely, the same backward with the table gradient
    exchanged; the table gradients must agree on every rank, otherwise the job falls back to the dense all-reduce.
    mode "auto" additionally TIMES every exchange that passed, and the dense all-reduce, and keeps the fastest.
    Returns the note that goes into the JSON line."""
    import time as _t

    import torch.nn.functional as F
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.distributed import GradAllReduce
    table = model.word_embeddings.embedding.weight
    tap = grad_sync.tap
    was_training = model.training
    model.eval()                                   # no dropout: both passes see the same forward
    candidates = {"auto": ["taps", "owner"], "taps": ["taps"], "owner": ["owner"]}[mode]
    passed, errs, desc_w = [], {}, None
    try:
        RF.set_tap_sink(None)
        model.zero_grad(set_to_none=True)
        F.mse_loss(model(*args), ratings).backward()
        GradAllReduce(model)(model)
        ref = table.grad.clone()
    except Exception as e:
        ref, errs["dense reference"] = None, repr(e)[:100]
    for m in candidates if ref is not None else []:
        ok = True
        try:
            tap.owner, tap.optimizer = (m == "owner"), None          # no optimizer: the exchange leaves a dense .grad to compare
            RF.set_tap_sink(tap)
            model.zero_grad(set_to_none=True)
            F.mse_loss(model(*args), ratings).backward()
            desc_w = (tap.desc, tap.weights)
            grad_sync(model)
            if m == "owner":
                tap.check()
            err = float((table.grad - ref).abs().max())
            ok = err <= 1e-7 + 2e-5 * float(ref.abs().max())
            errs[m] = f"{err:.1e}"
        except Exception as e:                      # any refusal of the path: not a candidate
            ok, errs[m] = False, repr(e)[:100]
        flag = torch.tensor([1 if ok else 0], device=table.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            passed.append(m)
    chosen, timing = (passed[0] if passed else None), ""
    if passed and mode == "auto":
        # which exchange is faster HERE depends on the links of this node: time them on the job's own ranks.  The tap
        # paths replace the local table-gradient chain (~0.1 ms), so they are charged their time minus that.
        def timed(fn, k=5):
            fn()
            dist.barrier(); torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize()
            return (_t.perf_counter() - t0) / k

        def once(m):
            def fn():
                tap.owner = (m == "owner")
                tap.desc, tap.weights, tap.calls = desc_w[0], desc_w[1], 1
                for h in tap.start():
                    h.wait()
                tap.finish()
            return fn

        dummy = torch.empty_like(table)
        op = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM
        ts = [timed(once(m)) for m in passed] + [timed(lambda: dist.all_reduce(dummy, op=op))]
        tt = torch.tensor(ts, dtype=torch.float64, device=table.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ts = [float(x) for x in tt]
        cost = {m: t - 1.0e-4 for m, t in zip(passed, ts)}
        cost["dense"] = ts[-1]
        chosen = min(cost, key=cost.get)
        timing = "; timed at start-up: " + ", ".join(f"{m} {t * 1e3:.3f} ms" for m, t in zip(passed + ["dense all-reduce"], ts))
        if chosen == "dense":
            chosen = None
    model.zero_grad(set_to_none=True)
    model.train(was_training)
    if chosen is not None:
        tap.owner = chosen == "owner"
        tap.optimizer = optimizer if tap.owner else None
        RF.set_tap_sink(tap)
        what = ("taps of the word-table gradient all-gathered" if chosen == "taps" else
                "taps of the word-table gradient all-gathered, rank r rebuilds the rows of the tokens t % N == r, the slabs are "
                "all-gathered and the optimizer reads them in row form")
        return (f"{what} (fp32, {tap.n * 8 / 1e6:.1f} MB of taps per rank instead of a {table.numel() * 4 / 1e6:.0f} MB all-reduce), "
                f"the other gradients all-reduced; checked against the dense all-reduce at start-up (max |diff| {errs[chosen]}){timing}")
    RF.set_tap_sink(None)
    grad_sync.tap = None
    if passed:
        return f"RCCL all-reduce of every gradient, fp32{timing}"
    return f"RCCL all-reduce of every gradient, fp32 (tap exchange failed its start-up check: {errs})"


def time_gemm_launches(model, args, reps=50):
    """HIP events on the launch stream around `reps` back-to-back launches of the conv's distinct-token GEMM
    (rbr_textcnn_prod_table) on a workspace prepared for this batch: the queue stays full, so the interval holds kernel
    time only.  Returns ms per launch."""
    import ctypes as C
    from review_based_recommender_amd import _lib
    L_ = _lib.lib()
    conv = model.ngram.feature_layer[0]
    table = model.word_embeddings.weight.detach()
    ids = torch.cat([args[0], args[1]]).contiguous()
    mask = torch.cat([args[2], args[3]]).contiguous().view(torch.uint8)
    ws_ = [w.detach().contiguous() for w in conv.weights()]
    V, D = table.shape
    desc = _lib.make_desc(ids.shape[0], ids.shape[1], D, V, [w.shape[2] for w in ws_], [w.shape[0] for w in ws_],
                          _lib.PAD_SAME, _lib.ACT_RELU, 0)
    nbytes = L_.rbr_textcnn_fwd_ws_bytes(C.byref(desc))
    if not nbytes:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=table.device)
    pidx = torch.empty(L_.rbr_textcnn_partial_elems(C.byref(desc)), dtype=torch.int32, device=table.device)
    st = _lib.current_stream()
    _lib.check(L_.rbr_textcnn_prod_prepare(C.byref(desc), ids.data_ptr(), mask.data_ptr(), _lib.ptr_array(ws_, torch.float32, "W"),
                                           pidx.data_ptr(), ws.data_ptr(), st), "prod_prepare")
    for _ in range(3):
        _lib.check(L_.rbr_textcnn_prod_table(C.byref(desc), table.data_ptr(), ws.data_ptr(), st), "prod_table")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(L_.rbr_textcnn_prod_table(C.byref(desc), table.data_ptr(), ws.data_ptr(), st), "prod_table")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_optimizer_launches(cfg, device, clipping=True, n_sets=3, reps=45):
    """In-step time of rbr_clip_adam_step[_rows] (its two kernels: gradient norm, clip + Adam): HIP events on the launch
    stream around `n_sets` back-to-back calls, one per ROTATING copy of the model + Adam state + gradients (3 x 210 MB: the
    working set exceeds the 256 MiB Infinity Cache, as it does inside a step, where the conv kernels have pushed p / m / v out
    since the last update).  Each set holds the gradients one backward left behind (the table's in compact row form); they are
    restored from a copy before every timed trio so that every call sees the same gradient.  `clipping`: whether the timed step's
    gradient norm exceeds max_grad_norm (the clip then re-writes every existing gradient element: 12 instead of 8 bytes each) --
    the calls here are given the same state.  Returns (ms per call, bytes per call as the kernels move them, ...)."""
    from review_based_recommender_amd.train_step import HipClipAdam, _forward_loss_backward, make_optimizer
    sets = []
    for k in range(n_sets):
        m = build_model(cfg, device)
        m.train()
        o = make_optimizer(m, hip_clip_adam=True)
        a, r = batch_on(cfg, 200 + k, device)
        o.zero_grad()
        _forward_loss_backward(m, a, r, o)
        rows = [(rg.rows, rg.rows.clone()) for rg in o._row_grads.values()] if isinstance(o, HipClipAdam) else []
        dense = [(p.grad, p.grad.clone()) for p in m.parameters() if p.grad is not None]
        sets.append((m, o, rows + dense))
    max_norm = 5.0 if clipping else 1e30
    for _, o, _ in sets:
        o.clip_and_step(max_norm)
    torch.cuda.synchronize()
    # the three calls recorded into ONE hipGraph: replayed, their six kernels run back to back (launched from Python a call costs
    # more host time than its kernels take on the GPU, and the event interval would hold the gaps)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _, o, _ in sets:
            o.clip_and_step(max_norm)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _, o, _ in sets:
            o.clip_and_step(max_norm)
    pairs = []
    for i in range(reps // n_sets):
        for _, _, saved in sets:           # every set clips as its first call did (restored outside the timed interval)
            for dst, src in saved:
                dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) / n_sets for a, b in pairs)
    m, o, _ = sets[0]
    n_par = sum(p.numel() for p in m.parameters())
    row_elems = sum(int(rg.rows.shape[1]) * n for rg, n in ((rg, _listed_rows(rg)) for rg in o._row_grads.values()))
    table_elems = sum(p.numel() for p in o._row_grads)
    # p, m, v read + written for every parameter; g read twice (norm, update) and -- when the clip scales it -- written once,
    # where it exists: dense gradients of the other tensors, the listed rows of the table (its other rows are never materialised)
    nbytes = 24.0 * n_par + (12.0 if clipping else 8.0) * ((n_par - table_elems) + row_elems)
    del sets
    torch.cuda.empty_cache()
    return ms[len(ms) // 2], nbytes, row_elems, table_elems


def _listed_rows(rg) -> int:
    """Rows of a compact row gradient that hold a token of the batch (counted on the device, read back once)."""
    return int(_row_of_token_tensor(rg).ge(0).sum().item())


def _row_of_token_tensor(rg):
    """row_of_token [V] of a RowGradient as a tensor (it lives inside the forward's workspace, which rg keeps alive)."""
    ws = rg._keep
    off = rg.row_of_token_ptr - ws.data_ptr()
    return ws[off:off + 4 * rg.V].view(torch.int32)


def dense_mode_env() -> bool:
    return os.environ.get("RBR_CONV_MODE") == "dense"



# --------------------------------------------------------------------------- the other BASELINE configs, in the same line
def _gpu_delay_calibration():
    """GPU milliseconds per million torch.cuda._sleep cycles (the spin kernel's clock differs between parts)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(4_000_000)
    e1.record()
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e1) / 4.0, 1e-4)


def queued_kernel_pass(run_step, n_steps, ms_per_mcycle):
    """HIP events around every C-ABI call (_lib.TIMER) of `n_steps` EAGER steps, each enqueued behind a spin kernel that holds the
    GPU back for longer than the host needs to enqueue the whole step: when the kernels run, the queue is full, so an event
    interval holds the kernel (and the cache state the step's own order leaves), not the host's launch gaps.  Calls that are
    ONE launch are therefore kernel times.  Returns ({call name: (calls, mean ms)}, host ms per eager step)."""
    from review_based_recommender_amd import _lib
    run_step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_step(1)
    host_ms = (time.perf_counter() - t0) * 1e3          # host time to enqueue one eager step
    torch.cuda.synchronize()
    cycles = int((1.5 * host_ms + 0.5) / ms_per_mcycle * 1e6)
    _lib.TIMER.start()
    for i in range(n_steps):
        torch.cuda._sleep(cycles)
        run_step(2 + i)
        torch.cuda.synchronize()
    _lib.TIMER.stop()
    return _lib.TIMER.summary(), host_ms


def price_kernel(name, ms, shape):
    """Roofline object of the C-ABI call `name` (one kernel launch; the optimizer: a launch pair) that took `ms`: algorithmic
    bytes / FLOPs of that launch (`shape`: distinct tokens, positions, columns of the product table, ...)
    over the HIP-event time."""
    nd, pos, cp, D, C_, tb = (shape[k] for k in ("distinct", "positions", "cp", "D", "C", "t_bytes"))
    slabs = shape["slabs"]

    def hbm(by, what):
        return {"bound": "hbm", "kernel": what, "bytes_per_launch": by, "achieved": round(by / ms / 1e6, 1), "peak": PEAK_HBM_GBPS,
                "unit": "GB/s", "frac": round(by / ms / 1e6 / PEAK_HBM_GBPS, 4)}

    if name == "textcnn_prod_pool":
        return hbm(nd * cp * tb + pos * (8 + 1 + 4 * shape["gates"]) + slabs * C_ * 8.0,
                   "gather_pool_kernel: per active 32-token slab, the sum of the kz product-table rows of every position, running max / "
                   "first argmax; bytes = every product-table row once + ids, masks, gates + the slab partials it writes")
    if name == "textcnn_bwd_g_product":
        return hbm(nd * cp * 4.0 + cp * D * 4.0 + nd * D * 4.0,
                   "g_times_w_kernel: word-table gradient rows = G[distinct tokens] @ Wprod^T as a sparse row product; bytes = G once + "
                   "Wprod^T once + one gradient row per distinct token")
    if name == "clip_adam_step":
        return hbm(24.0 * shape["n_par"] + 12.0 * shape["grad_elems"],
                   "grad_sqnorm_kernel + clip_adam_kernel (rbr_clip_adam_step[_rows]: clip_grad_norm_ + Adam over every parameter "
                   "tensor; a launch PAIR); bytes = p, m, v read + written + every existing gradient element read twice, written once")
    if name == "textcnn_prod_table":
        nprod = shape["nprod"]
        fl = 2.0 * nd * D * cp
        f32 = shape["precision"] == "f32"
        peak = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_BF16_MFMA_TFLOPS
        what = ("prod_gemm_kernel: T = table[distinct tokens] @ Wprod on v_mfma_f32_32x32x2_f32" if f32 else
                f"prod_gemm_b16* kernel: T = table[distinct tokens] @ Wprod, {nprod} bf16 plane product(s) per f32 product on "
                "v_mfma_f32_32x32x16_bf16")
        return {"bound": "mfma", "kernel": what, "flops_per_launch": nprod * fl, "algorithmic_flops_per_launch": fl,
                "achieved": round(nprod * fl / ms / 1e9, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(nprod * fl / ms / 1e9 / peak, 4)}
    return None


SINGLE_KERNEL_CALLS = ("textcnn_prod_pool", "textcnn_bwd_g_product", "textcnn_prod_table", "clip_adam_step")


def secondary_configs(device, a, precision0):
    """BASELINE.json configs[1] in the bf16 class, configs[2] (NARRE cfg3: bf16 class -- the dtype BASELINE names -- and f32
    class) and configs[3] (D-ATT cfg4): the trainer step of each (dropout 0.5, clip, Adam; GraphedTrainStep + HipClipAdam, one
    recorded graph per resident batch, as the headline), timed the headline's way, with the kernel launches of a replay, the
    step's longest single kernel priced against its roofline (HIP events, queued_kernel_pass) and max |pred - fixture| on the
    config's golden batch (tests/golden/*.npz: the reference's own forward) so every number carries its parity class."""
    import contextlib
    import io

    import numpy as np

    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step

    def quiet(fn, *args):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(*args)

    def on_dev(b, keys):
        return tuple(b[k].to(device) for k in keys), b["ratings"].to(device)

    def deepconn(cfg):
        m = build_model(cfg, device)
        keys = ("u_docs", "i_docs", "u_masks", "i_masks", "u_ids", "i_ids")
        return (m, lambda sd: on_dev(synth.deepconn_batch(cfg, sd), keys),
                lambda args: (torch.cat([args[0], args[1]]), torch.cat([args[2], args[3]])),
                dict(kz=cfg["kz"], ch=[cfg["H"] // len(cfg["kz"])] * len(cfg["kz"]), D=cfg["D"], gates=0))

    def narre(cfg):
        from review_based_recommender_amd.models.narre.narre import NARRE
        m = quiet(NARRE, cfg["U"], cfg["I"], cfg["V"], cfg["kz"], cfg["H"], cfg["D"], cfg["A"], cfg["K"], cfg["R"], cfg["T"], 0.5,
                  0, 0, 0, None, "CNN")
        m.load_state_dict(synth.narre_params(cfg, 0))
        keys = ("u_text", "i_text", "u_masks", "i_masks", "u_id", "i_id", "reuid", "reiid")
        return (m.to(device), lambda sd: on_dev(synth.narre_batch(cfg, sd), keys),
                lambda args: (torch.cat([args[0], args[1]]).view(-1, cfg["T"]), torch.cat([args[2], args[3]]).view(-1, cfg["T"])),
                dict(kz=cfg["kz"], ch=[cfg["H"] // len(cfg["kz"])] * len(cfg["kz"]), D=cfg["D"], gates=0))

    def datt(cfg):
        from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
        m = quiet(DualAtt, cfg["V"], cfg["L"], cfg["win"], cfg["l_out"], cfg["g_out"], cfg["E"], cfg["h1"], cfg["h2"], 0.5, None)
        m.load_state_dict(synth.datt_params(cfg, 0, table_scale=0.3))          # as the fixture
        return (m.to(device), lambda sd: on_dev(synth.datt_batch(cfg, sd), ("u_docs", "i_docs")),
                lambda args: (torch.cat([args[0], args[1]]), None),
                dict(kz=[1, 2, 3, 4], ch=[cfg["l_out"]] + [cfg["g_out"]] * 3, D=cfg["E"], gates=1))

    specs = [
        ("deepconn_cfg2_bf16", "DeepCoNN cfg2 (BASELINE configs[1]'s shape) in the bf16 class: B=256, 2x512 tokens, D=300, widths 3/5/7 x 50",
         deepconn, synth.DEEPCONN_CFGS["cfg2"], "bf16", "deepconn_cfg2", 256),
        ("narre_cfg3_bf16", "NARRE cfg3 (BASELINE configs[2]): B=256, 10 reviews x 50 tokens per side, D=300, width 3 x 150, bf16 class",
         narre, synth.NARRE_CFGS["cfg3"], "bf16", "narre_cfg3", 256),
        ("narre_cfg3_f32", "NARRE cfg3, f32 class (bf16x3: exact three-plane split)", narre, synth.NARRE_CFGS["cfg3"], None,
         "narre_cfg3", 256),
        ("datt_cfg4_f32", "D-ATT cfg4 (BASELINE configs[3]): B=512, 2x1024 tokens, E=100, local 200 + global 3x100 channels, f32 class",
         datt, synth.DATT_CFGS["cfg4"], None, "datt_cfg4", 512),
    ]
    ms_per_mcycle = _gpu_delay_calibration()
    out = {}
    for key, label, build, cfg, prec, fixture, B in specs:
        t_cfg = time.perf_counter()
        try:
            RF.set_prod_precision(prec)
            cls = RF.get_prod_precision()
            model, bat, tok, conv = build(cfg)
            # ---- parity class of this configuration: eval forward on the golden batch against the reference's recording
            g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
            gargs, _ = bat(1)
            model.eval()
            with torch.no_grad():
                o = model(*gargs)
                pred = (o[0] if isinstance(o, tuple) else o).float().cpu().numpy()
            err = float(np.abs(pred.astype(np.float64) - g["pred_eval"].astype(np.float64)).max())
            tol = 3e-2 if cls == "bf16" else 1e-4
            # ---- the timed step
            model.train()
            opt = make_optimizer(model, hip_clip_adam=True)
            nb = max(1, a.batches)
            bs = [bat(1 + j) for j in range(nb)]
            with torch.no_grad():
                init = [p.detach().clone() for p in model.parameters()]

            def reset():
                with torch.no_grad():
                    for p, v in zip(model.parameters(), init):
                        p.copy_(v)
                    for st_ in opt.state.values():
                        for v in st_.values():
                            if torch.is_tensor(v):
                                v.zero_()
                torch.cuda.synchronize()

            stepper = GraphedTrainStep(model, opt, bs[0][0], bs[0][1], slots=nb, keep_graph=True)
            launches = stepper.kernel_launches()
            for k, b in enumerate(bs):
                stepper.stage(k, *b)
            reset()
            for i in range(a.warmup):
                stepper(slot=i % nb)
            blocks, k = [], 0
            while True:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(a.steps):
                    stepper(slot=(k + i) % nb)
                torch.cuda.synchronize()
                blocks.append(time.perf_counter() - t0)
                k += a.steps
                if len(blocks) >= 3 and sum(blocks) >= 0.25:
                    break
            med = statistics.median(blocks)
            # ---- kernel pass: the same step launched eagerly behind a spin kernel (full queue), HIP events per C-ABI call
            reset()
            n_pass = 6
            ksum, host_ms = queued_kernel_pass(lambda i: train_step(model, opt, bs[i % nb][0], bs[i % nb][1]), n_pass, ms_per_mcycle)
            ids, msk = tok(bs[0][0])
            live = ids[msk] if msk is not None else ids.reshape(-1)
            n_par = sum(p.numel() for p in model.parameters())
            table = next(p for n_, p in model.named_parameters() if n_.startswith("word_embeddings"))
            distinct = int(torch.unique(live).numel())
            row_form = len(getattr(opt, "_row_grads", {})) > 0
            shape = dict(distinct=distinct, positions=int(ids.numel()), cp=sum(k_ * c_ for k_, c_ in zip(conv["kz"], conv["ch"])),
                         D=conv["D"], C=sum(conv["ch"]), t_bytes=2 if cls == "bf16" else 4,
                         slabs=int(ids.shape[0]) * ((int(ids.shape[1]) + 31) // 32), gates=conv["gates"],
                         nprod={"f32": 1, "bf16x3": 6, "bf16x2": 3, "bf16": 1}[cls], precision=cls, n_par=n_par,
                         grad_elems=(n_par - table.numel() + distinct * conv["D"]) if row_form else n_par)
            single = {n_: v for n_, v in ksum.items() if n_ in SINGLE_KERNEL_CALLS}
            dom = max(single, key=lambda n_: single[n_][1]) if single else None
            roof = None
            if dom is not None:
                roof = price_kernel(dom, single[dom][1], shape)
                roof.update({"call": dom, "avg_launch_ms": round(single[dom][1], 4), "launches_timed": single[dom][0],
                             "calls_per_step": round(single[dom][0] / float(n_pass), 2),
                             "timing": "HIP events around the C-ABI call in eager steps enqueued behind a spin kernel (full queue)",
                             "traffic": None})
            out[key] = {
                "workload": label, "ms_per_step": round(1e3 * med / a.steps, 4), "ms_per_step_min": round(1e3 * min(blocks) / a.steps, 4),
                "pairs_per_s": round(B * a.steps / med, 1), "repeats": len(blocks), "launches_per_step": launches,
                "launch": f"hipGraph replay, {nb} resident batches, one recorded step per input slot", "conv_arithmetic": cls,
                "max_abs_pred_err_vs_fixture": float(f"{err:.3e}"), "fixture": f"tests/golden/{fixture}.npz (pred_eval: the reference's forward)",
                "parity_tolerance": tol, "parity_ok": bool(err <= tol), "distinct_tokens": distinct, "positions": shape["positions"],
                "roofline": roof,
                "kernels_ms": {n_: round(v[1] * v[0] / float(n_pass), 4)
                               for n_, v in sorted(ksum.items(), key=lambda kv: -kv[1][1] * kv[1][0])[:8]},
                "kernels_ms_note": "per step: mean HIP-event ms of the call x calls per step (a multi-launch call holds all its launches)",
                "eager_host_ms_per_step": round(host_ms, 3), "wall_s": None}
            del stepper, opt, model, bs, init
        except Exception as e:          # one configuration must not cost the line
            out[key] = {"workload": label, "error": f"{type(e).__name__}: {str(e)[:200]}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if "wall_s" in out[key]:
            out[key]["wall_s"] = round(time.perf_counter() - t_cfg, 1)
    RF.set_prod_precision(None if precision0 == "bf16x3" else precision0)
    return out


def self_launch(a) -> int:
    """--gpus N > 1 from a plain invocation: start the N ranks as a child torch.distributed.run (this process has
    not touched the GPU, and never does), relay its output and return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["RBR_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches rotated through the step's input block")
    ap.add_argument("--one-slot", action="store_true",
                    help="record ONE step graph and copy every batch into its input block (the copy_per_step form) instead of one graph per resident batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the single-batch / uniform-id / all-distinct secondary results")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16x2", "bf16"], default=None,
                    help="arithmetic of the token-product GEMM (default bf16x3: exact three-plane split, f32-class accuracy)")
    ap.add_argument("--comm-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="wire format of the word-table gradient all-reduce (N > 1); fp32 is exact")
    ap.add_argument("--exchange", choices=["auto", "taps", "owner", "dense"], default="auto",
                    help="N > 1, word-table gradient: all-gather its taps (4.3 MB per rank) and rebuild the gradient on every rank "
                         "(taps) or by owner (owner: rank r builds the rows of tokens t %% N == r, slabs all-gathered), or all-reduce "
                         "the dense 60 MB gradient; auto times all three on the job's own ranks at start-up and keeps the fastest")
    ap.add_argument("--torch-optim", action="store_true",
                    help="clip_grad_norm_ + torch.optim.Adam(fused) instead of the two-launch HipClipAdam")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` object (cfg2 in the bf16 class, NARRE cfg3 in both classes, D-ATT cfg4)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))          # nothing above this line has initialised HIP

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    from review_based_recommender_amd import _lib
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step
    _lib.lib()   # fail loudly now if librbr_hip.so is missing
    if a.precision is not None:
        RF.set_prod_precision(a.precision)
    precision = RF.get_prod_precision()

    # rehearsal aids for a one-GPU box (never set by the driver): all ranks on device 0, gloo instead of RCCL
    if os.environ.get("RBR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("RBR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = synth.DEEPCONN_CFGS[WORKLOAD]

    grad_sync = None
    exchange_note = None
    dist = None
    if world > 1:
        import torch.distributed as dist
        from review_based_recommender_amd.distributed import GradAllReduce, init_process_group_from_env
        init_process_group_from_env(backend)

    model = build_model(cfg, device)          # identical parameters on every rank (same seed)
    model.train()
    use_graph = not a.no_graph
    opt = make_optimizer(model, capturable=use_graph, hip_clip_adam=not a.torch_optim)
    nb = max(1, a.batches)
    seeds = [1 + rank * 64 + j for j in range(nb)]      # each rank owns different shards
    batches = [batch_on(cfg, sd, device) for sd in seeds]
    args, ratings = batches[0]
    if world > 1:
        use_taps = a.comm_dtype == "fp32" and a.exchange in ("taps", "owner", "auto")
        grad_sync = GradAllReduce(model, comm_dtype=torch.bfloat16 if a.comm_dtype == "bf16" else None,
                                  tap_table=model.word_embeddings.embedding.weight if use_taps else None)
        if use_taps:
            exchange_note = verify_tap_exchange(model, args, ratings, grad_sync, dist, mode=a.exchange,
                                                optimizer=opt if not a.torch_optim else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    launch_note = None
    stepper = None
    if use_graph:
        try:
            stepper = GraphedTrainStep(model, opt, args, ratings, grad_sync=grad_sync, slots=1 if a.one_slot else len(batches))
        except Exception as e:      # a capture the runtime refuses must not cost the measurement: launch eagerly instead
            use_graph = False
            launch_note = f"hipGraph capture failed ({type(e).__name__}: {str(e)[:120]}); eager launches"
            torch.cuda.synchronize()

    def make_runner(bs, st=None, sync=None, copy_per_step=False):
        """step(i) over the resident batches `bs`: batch k lives in input slot k of the step (staged here, outside the timed
        region); with fewer slots than batches -- or copy_per_step -- every step copies its batch into slot 0 first."""
        st = stepper if st is None else st
        sync = grad_sync if sync is None else sync
        if use_graph:
            if len(bs) <= st.slots and not copy_per_step:
                for k, b in enumerate(bs):
                    st.stage(k, *b)      # resident from here on
                torch.cuda.synchronize()
                n = len(bs)
                return lambda i: st(slot=i % n)
            blobs = [st.pack(*b) for b in bs]
            return lambda i: st(packed=blobs[i % len(blobs)])
        return lambda i: train_step(model, opt, bs[i % len(bs)][0], bs[i % len(bs)][1], grad_sync=sync)

    def timed_blocks(step, min_s=MIN_TIMED_S, max_repeats=400):
        """Blocks of exactly a.steps steps, barrier + synchronize on both sides, max over ranks; repeated until the timed
        region adds up to min_s.  Returns the per-block seconds."""
        for i in range(a.warmup):
            step(i)
        out = []
        k = 0
        target = None
        while True:
            barrier()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(k + i)
            barrier()
            out.append(max_over_ranks(time.perf_counter() - t0))
            k += a.steps
            if target is None:      # the same count on every rank: derived from the all-reduced first block
                target = min(max_repeats, max(3, int(min_s / max(out[0], 1e-6)) + 1))
            if len(out) >= target:
                return out

    # Every timed measurement starts from the SAME model: the step gets cheaper as the repeated synthetic batches are overfitted
    # (dead ReLUs carry no taps), so a measurement taken after another one would be flattered.
    with torch.no_grad():
        init_params = [p.detach().clone() for p in model.parameters()]

    def reset_model():
        """Parameters back to their initial values, Adam state to zero (in place: recorded graphs keep their addresses)."""
        with torch.no_grad():
            for p, v in zip(model.parameters(), init_params):
                p.copy_(v)
            for st_ in opt.state.values():
                for v in st_.values():
                    if torch.is_tensor(v):
                        v.zero_()
        torch.cuda.synchronize()

    reset_model()
    headline = timed_blocks(make_runner(batches), max_repeats=2000)
    med, best = statistics.median(headline), min(headline)
    if os.environ.get("RBR_BENCH_DRIFT") and rank == 0:
        q = max(1, len(headline) // 8)
        print("drift (ms/step, eighths of the timed region):",
              [round(1e3 * statistics.median(headline[k * q:(k + 1) * q]) / a.steps, 4) for k in range(8)], file=sys.stderr, flush=True)

    variants = {}
    if not a.no_variants:
        reset_model()
        one = timed_blocks(make_runner(batches[:1]), min_s=0.25)
        variants["single_batch_replay"] = {"ms_per_step": round(1e3 * statistics.median(one) / a.steps, 4),
                                           "ms_per_step_min": round(1e3 * min(one) / a.steps, 4), "repeats": len(one),
                                           "note": "one batch resident in the step's input block, no per-step copy (round 1's line)"}
        if use_graph and stepper.slots > 1:
            reset_model()
            cp = timed_blocks(make_runner(batches, copy_per_step=True), min_s=0.25)
            variants["copy_per_step"] = {"ms_per_step": round(1e3 * statistics.median(cp) / a.steps, 4),
                                         "ms_per_step_min": round(1e3 * min(cp) / a.steps, 4), "repeats": len(cp),
                                         "note": "one input slot: every step copies its batch device-to-device into the graph's "
                                                 "input block inside the timed region (the headline of rounds 2 and 3)"}
        for kind in ("uniform_ids", "all_distinct"):
            vb = [batch_on(cfg, sd, device, ids=kind.replace("_ids", "")) for sd in seeds]
            reset_model()
            blk = timed_blocks(make_runner(vb), min_s=0.25)
            ids_all = torch.cat([vb[0][0][0], vb[0][0][1]])
            msk = torch.cat([vb[0][0][2], vb[0][0][3]])
            variants[kind] = {"ms_per_step": round(1e3 * statistics.median(blk) / a.steps, 4),
                              "pairs_per_s": round(cfg["B"] * world * a.steps / statistics.median(blk), 1),
                              "repeats": len(blk), "distinct_tokens": int(torch.unique(ids_all[msk]).numel()),
                              "positions": int(msk.numel())}
            del vb

    # HIP-event pass over the same K steps of the headline batches, launched eagerly
    run_eager = lambda i: train_step(model, opt, batches[i % nb][0], batches[i % nb][1], grad_sync=grad_sync)
    for i in range(3):
        run_eager(i)
    barrier()
    _lib.TIMER.start()
    for i in range(a.steps):
        run_eager(i)
    barrier()
    _lib.TIMER.stop()
    ksum = _lib.TIMER.summary()

    gemm_ms = opt_ms = None
    opt_bytes = opt_rows = opt_table = 0
    step_clips = True
    if rank == 0 and not dense_mode_env():
        gemm_ms = time_gemm_launches(model, args)
        if not a.torch_optim:
            # the clip's state in the timed steps (their last gradient norm): the timing loop is given the same
            step_clips = True if stepper is None else bool(float(stepper.gnorm) > 5.0)
            opt_ms, opt_bytes, opt_rows, opt_table = time_optimizer_launches(cfg, device, clipping=step_clips)

    # forward-only (eval) rate, reported beside the headline
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(*args)
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        for _ in range(20):
            model(*args)
        torch.cuda.synchronize()
        fwd_s = (time.perf_counter() - f0) / 20
    fwd_graph_s = None
    if use_graph:                                   # the same forward replayed from a hipGraph (train_step.GraphedForward)
        try:
            from review_based_recommender_amd.train_step import GraphedForward
            gf = GraphedForward(model, args)
            blobs = [gf.pack(b[0]) for b in batches]
            for i in range(5):
                gf(packed=blobs[i % nb])
            torch.cuda.synchronize()
            f0 = time.perf_counter()
            for i in range(100):
                gf(packed=blobs[i % nb])
            torch.cuda.synchronize()
            fwd_graph_s = (time.perf_counter() - f0) / 100
        except Exception as e:
            launch_note = (launch_note or "") + f" [forward graph failed: {type(e).__name__}]"

    # BASELINE configs[4] names this job's 8-GPU point in bf16 ("batch 2048 data-parallel, RCCL grad all-reduce, bf16"): the same
    # ranks once more with the conv contraction in plain bf16 (f32 accumulate) and every gradient all-reduced densely in a bf16
    # wire format.  A variant beside the fp32 line (whose dtype matches the N = 1 point), never the headline.
    if world > 1 and not a.no_variants:
        grad_sync.close()                               # uninstalls the tap sink: the table gradient is dense again
        RF.set_prod_precision("bf16")
        model.train()
        sync5 = GradAllReduce(model, comm_dtype=torch.bfloat16)
        try:
            st5 = GraphedTrainStep(model, opt, args, ratings, grad_sync=sync5, slots=stepper.slots) if use_graph else None
            reset_model()
            blk = timed_blocks(make_runner(batches, st=st5, sync=sync5), min_s=0.25)
            variants["cfg5_bf16"] = {"ms_per_step": round(1e3 * statistics.median(blk) / a.steps, 4),
                                     "pairs_per_s": round(cfg["B"] * world * a.steps / statistics.median(blk), 1),
                                     "repeats": len(blk), "dtype": "bf16 conv operands (f32 accumulate), f32 parameters and Adam state",
                                     "grad_exchange": "RCCL all-reduce of every gradient, bf16 wire format"}
        except Exception as e:
            variants["cfg5_bf16"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
        RF.set_prod_precision(precision)

    if rank == 0:
        flops = conv_fwd_flops(cfg)
        masks = torch.cat([args[2], args[3]])
        dense_mode = "textcnn_conv_fwd" in ksum
        out = {
            "metric": "(user,item) pairs/sec, DeepCoNN train step (fwd+MSE+bwd+clip+Adam), bsz256 2x512tok",
            "value": round(cfg["B"] * world * a.steps / med, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * med / a.steps, 4), "ms_per_step_min": round(1e3 * best / a.steps, 4),
            "repeats": len(headline), "timed_region_s": round(sum(headline), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if precision == "bf16" else "f32", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
            "config": {"workload": "DeepCoNN cfg2: batch 256 pairs/GPU, 2x512-token docs, D=300, conv widths 3/5/7 x 50, "
                                   "latent 32, V=50002, fp32, Zipf ids", "global_batch": cfg["B"] * world,
                       "parallelism": f"dp{world}", "launch": launch_note or (("hipGraph replay, one recorded step per resident input slot (no per-step input copy)"
                                                  if stepper.slots > 1 else "hipGraph replay") if use_graph else "eager"),
                       "resident_batches": nb,
                       "conv_arithmetic": {"f32": "f32 MFMA (v_mfma_f32_32x32x2_f32)",
                                           "bf16x3": "f32 operands split exactly into 3 bf16 planes, 6 plane products per product on "
                                                     "v_mfma_f32_32x32x16_bf16, f32 accumulate (f32-class accuracy: parity tests at 1e-4)",
                                           "bf16x2": "2 bf16 planes, 3 plane products, f32 accumulate",
                                           "bf16": "operands rounded to bf16, f32 accumulate (reduced precision)"}[precision],
                       "optimizer": "torch clip_grad_norm_ + fused Adam" if a.torch_optim else
                                    "HipClipAdam (clip + Adam, 2 launches; word-table gradient in compact row form)",
                       "grad_exchange": None if world == 1 else (exchange_note or f"RCCL all-reduce, {a.comm_dtype} wire format, before the clip")},
            "fwd_only_pairs_per_s": round(cfg["B"] / fwd_s, 1),
            "fwd_only_graph_pairs_per_s": None if fwd_graph_s is None else round(cfg["B"] / fwd_graph_s, 1),
            "kernels_ms": {k: round(v[1], 4) for k, v in ksum.items()},
            "kernel_timing": "HIP events around the C-ABI launches, eager pass over the same steps after the timed region",
        }
        out.update(variants)
        ids_all = torch.cat([args[0], args[1]])
        n_distinct = int(torch.unique(ids_all[masks]).numel())
        if dense_mode:
            conv_calls, conv_ms = ksum["textcnn_conv_fwd"]
            act_frac = active_tile_fraction(cfg, masks)
            ach = flops / (conv_ms * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "mfma", "kernel": "conv_fwd_kernel (dense: gather+conv+max-pool, v_mfma_f32_32x32x2_f32)",
                "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": measured_traffic("r01_conv_fwd_pmc.json"),
                "traffic_source": "profiles/r01_conv_fwd_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                "flops_per_launch": flops, "avg_launch_ms": round(conv_ms, 4), "launches_timed": conv_calls,
                "note": "achieved = ALGORITHMIC dense-conv FLOPs / time. Wave-tiles whose tokens are all padding are "
                        "skipped (their sums are exactly 0): executed_* price only the MFMAs issued",
                "executed_tile_fraction": round(act_frac, 4), "executed_achieved": round(ach * act_frac, 2),
                "executed_frac": round(ach * act_frac / PEAK_F32_MFMA_TFLOPS, 4)}
        else:
            # The conv runs in its token-product form; its contraction is ONE launch of the MFMA kernel over the distinct
            # tokens of the batch.  Algorithmic FLOPs of that formulation: 2 * distinct tokens * D * sum(kz * channels)
            # (DESIGN.md section 4); the bf16-plane arithmetics execute 6 / 3 / 1 bf16 MFMA products per f32 product.
            gemm_calls = 50
            if gemm_ms is None:
                gemm_calls, gemm_ms = ksum["textcnn_prod_table"]
            cp = sum(k * (cfg["H"] // len(cfg["kz"])) for k in cfg["kz"])
            gemm_flops = 2.0 * n_distinct * cfg["D"] * cp
            nprod = {"f32": 1, "bf16x3": 6, "bf16x2": 3, "bf16": 1}[precision]
            peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_BF16_MFMA_TFLOPS
            # the prepare stage of the b16 arithmetics adds the weight-plane pack launch in front of the GEMM; the events
            # bracket rbr_textcnn_prod_table alone
            ach = nprod * gemm_flops / (gemm_ms * 1e-3) / 1e12
            gemm_pmc = "r03_prod_gemm_b16_pmc.json" if precision != "f32" else "r01_prod_table_pmc.json"
            out["roofline_gemm"] = {
                "bound": "mfma",
                "kernel": ("prod_gemm_kernel<60> (v_mfma_f32_32x32x2_f32)" if precision == "f32" else
                           "prod_gemm_b16sd_kernel (bf16 storage: rows rounded in registers, v_mfma_f32_32x32x16_bf16, bf16 product table)" if (precision == "bf16" and os.environ.get("RBR_B16_STORAGE", "1") != "0") else
                           f"prod_gemm_b16d_kernel<{nprod}, 8> (v_mfma_f32_32x32x16_bf16, {nprod} plane products per f32 product; token "
                           f"rows loaded into registers, weight planes through LDS)")
                          + ": T = table[distinct tokens] @ Wprod",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": measured_traffic(gemm_pmc) or measured_traffic("r02_prod_gemm_b16_pmc.json"),
                "traffic_source": f"profiles/{gemm_pmc} (r02_ when this round's pass is absent)",
                "algorithmic_flops_per_launch": gemm_flops, "executed_mfma_flops_per_launch": nprod * gemm_flops,
                "avg_launch_ms": round(gemm_ms, 4), "launches_timed": gemm_calls,
                "algorithmic_vs_f32_mfma_peak": round(gemm_flops / (gemm_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "distinct_tokens": n_distinct, "positions": int(masks.numel()),
                "timing": "HIP events on the launch stream around 50 back-to-back launches on a workspace prepared for this batch",
                "note": "achieved = plane-product FLOPs the kernel executes on the bf16 pipe / HIP-event time; "
                        "algorithmic_vs_f32_mfma_peak = the f32 products it stands for / time / 157.3 TF (the pipe round 1 used)"}
            conv_ms = sum(ksum[k][1] for k in ("textcnn_prod_prepare", "textcnn_prod_table", "textcnn_prod_pool"))
            out["conv_stage"] = {
                "formulation": "token-product: T = table[distinct tokens] @ W on the MFMA pipe, then gather-add + max-pool",
                "ms": round(conv_ms, 4), "dense_conv_flops_it_replaces": flops}
            if opt_ms is not None:
                # longest kernel pair of the step: clip_grad_norm_ + Adam over all 17 tensors.  Algorithmic bytes per call
                # (DESIGN.md section 4): p, m, v read and written for every parameter (24 B each); the gradient read for the
                # norm, read for the update and written back clipped (12 B) where it EXISTS -- the dense gradients of the small
                # tensors and the rows of the batch's tokens in the word table's compact row gradient.
                n_par = sum(p.numel() for p in model.parameters())
                gbps = opt_bytes / (opt_ms * 1e-3) / 1e9
                out["roofline"] = {
                    "bound": "hbm", "kernel": "grad_sqnorm_kernel + clip_adam_kernel<1> (rbr_clip_adam_step_rows: clip_grad_norm_ + Adam, "
                                              "all 17 parameter tensors, the word table's gradient in compact row form)",
                    "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                    "traffic": (lambda a_, b_: None if a_ is None or b_ is None else a_ + b_)(
                        measured_traffic("r03_clip_adam_pmc.json"), measured_traffic("r03_grad_sqnorm_pmc.json")),
                    "traffic_source": "profiles/r03_clip_adam_pmc.json + r03_grad_sqnorm_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes, gfx950 correction)",
                    "bytes_per_launch": opt_bytes, "avg_launch_ms": round(opt_ms, 4), "launches_timed": 45,
                    "timing": "HIP events on the launch stream around a replayed hipGraph of 3 rbr_clip_adam_step_rows calls (two kernels "
                              "each), one per rotating copy of parameters + Adam state + gradients (630 MB: beyond the 256 MiB "
                              "Infinity Cache, as inside a step), median of 15 replays / 3; compare clip_adam_kernel<1> + "
                              "grad_sqnorm_kernel in profiles/r03_bench_kernel_stats.csv",
                    "clip_active": step_clips,
                    "note": f"24 B x {n_par} parameters (p, m, v read + written) + {12 if step_clips else 8} B x ({n_par - opt_table} "
                            f"dense-gradient elements + {opt_rows} elements of listed table rows: norm read, update read"
                            f"{', clipped write-back' if step_clips else '; the timed steps do not clip, so no write-back'}); per pair: "
                            f"{opt_bytes / cfg['B'] / 1e6:.2f} MB of optimizer traffic"}
            else:
                out["roofline"] = out["roofline_gemm"]
        if "roofline" in out and not dense_mode:
            # the whole step against HBM: unique bytes every stage has to move at least once, 8 TB/s
            n_par = sum(p.numel() for p in model.parameters())
            D, V = cfg["D"], cfg["V"]
            cp = sum(k * (cfg["H"] // len(cfg["kz"])) for k in cfg["kz"])
            n_pos = int(masks.numel())
            parts = {
                "optimizer (p, m, v r+w; existing gradients r, r[, w when clipping])":
                    24.0 * n_par + (12.0 if step_clips else 8.0) * ((n_par - V * D) + n_distinct * D),
                "distinct-token GEMM (table rows in, product table T out)": 4.0 * n_distinct * (D + 768),
                "gather + max-pool (T rows once, ids + masks)": 4.0 * n_distinct * 768 + 9.0 * n_pos,
                "table gradient (G zero + read, compact rows out)": 4.0 * n_distinct * (2 * (cp + 2) + D),
                "conv weight gradient (table rows once)": 4.0 * n_distinct * D,
            }
            step_bytes = sum(parts.values())
            floor_ms = step_bytes / (PEAK_HBM_GBPS * 1e9) * 1e3
            out["roofline"]["step"] = {"algorithmic_bytes": step_bytes, "floor_ms": round(floor_ms, 4),
                                       "frac": round(floor_ms / (1e3 * med / a.steps), 4), "ms_per_step": round(1e3 * med / a.steps, 4),
                                       "parts_bytes": {k: round(v) for k, v in parts.items()},
                                       "note": "unique HBM bytes of a step / 8 TB/s over the measured step; the conv-stage kernels are "
                                               "bound by L2 requests (rows re-read from L2 / Infinity Cache), not by these bytes"}
        if world == 1 and not a.no_configs:
            stepper = None                      # the headline's graphs and their memory pools are not needed any more
            torch.cuda.empty_cache()
            try:
                out["configs"] = secondary_configs(device, a, precision)
            except Exception as e:          # the secondary configurations must not cost the headline's line
                out["configs"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_budget)
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
