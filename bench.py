#!/usr/bin/env python3
"""bench.py -- (user,item) pairs/s of the DeepCoNN review-encoder train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] ("cfg2"): DeepCoNN, batch 256 pairs per GPU, 2 x 512-token
documents per pair, 300-d embeddings, conv widths 3/5/7 x 50 channels, latent 32, V = 50 002,
fp32, synthetic Zipf token ids (tests/golden/synth.py).  One step = the reference trainer's step
(trainer/train_deepconn_pp.py:161-168): zero_grad -> forward -> MSELoss -> backward ->
[N>1: RCCL gradient all-reduce] -> clip_grad_norm_(5.0) -> Adam(lr=2e-3).  Inputs are resident in
HBM before the timed region.  Weak scaling: every rank processes its own 256 pairs.

The step is recorded once into a hipGraph (train_step.GraphedTrainStep) and replayed on the batch resident in the
graph's input buffers.  `--no-graph` launches the kernels one by one from Python instead.

Rank 0 prints ONE JSON line.  `value` = pairs/s over all ranks of the full train step;
`roofline` prices the longest kernel of the step -- by default the distinct-token GEMM of the token-product
conv (csrc/textcnn_prod.hip) on the f32 MFMA pipe; with RBR_CONV_MODE=dense the fused gather+conv+max-pool
kernel -- with HIP events recorded around its launches on the launch stream.  Events cannot be read inside a
replayed graph, so the event pass re-runs the same K steps eagerly right after the timed region (with
--no-graph the events sit in the timed region itself).  `cpu_baseline` times the CPU oracle
(oracle/ref_cpu.py, torch CPU ops) on the same workload on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

import synth  # noqa: E402

WORKLOAD = "cfg2"
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


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


def batch_on(cfg, seed, device):
    b = synth.deepconn_batch(cfg, seed)
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


def verify_tap_exchange(model, args, ratings, grad_sync, dist, choose_by_time=False):
    """Start-up self-check of the tap exchange (distributed.TapExchange) on this job's own ranks: one backward whose
    gradients are all-reduced densely, the same backward with the table gradient exchanged as taps; the two table
    gradients must agree on every rank, otherwise the job falls back to the dense all-reduce.  Returns the note that goes
    into the JSON line."""
    import torch.nn.functional as F
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.distributed import GradAllReduce
    table = model.word_embeddings.embedding.weight
    was_training = model.training
    model.eval()                                   # no dropout: both passes see the same forward
    ok = True
    tap_desc = tap_weights = None
    try:
        RF.set_tap_sink(None)
        model.zero_grad(set_to_none=True)
        F.mse_loss(model(*args), ratings).backward()
        GradAllReduce(model)(model)
        ref = table.grad.clone()
        RF.set_tap_sink(grad_sync.tap)
        model.zero_grad(set_to_none=True)
        F.mse_loss(model(*args), ratings).backward()
        tap_desc, tap_weights = grad_sync.tap.desc, grad_sync.tap.weights
        grad_sync(model)
        err = float((table.grad - ref).abs().max())
        ok = err <= 1e-7 + 2e-5 * float(ref.abs().max())
    except Exception as e:                          # any refusal of the path: dense exchange
        ok, err = False, repr(e)[:100]
    flag = torch.tensor([1 if ok else 0], device=table.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = int(flag.item()) == 1
    note = None
    if ok and choose_by_time:
        # which exchange is faster HERE depends on the links of this node: time both on the job's own ranks.  The tap
        # path replaces the local table-gradient chain (~0.1 ms), so it is charged its time minus that.
        import time as _t
        tap = grad_sync.tap

        def timed(fn, k=5):
            fn()
            dist.barrier(); torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize()
            return (_t.perf_counter() - t0) / k

        def taps_once():
            tap.desc, tap.weights, tap.calls = desc_w[0], desc_w[1], 1
            for h in tap.start():
                h.wait()
            tap.finish()

        desc_w = (tap_desc, tap_weights)
        dummy = torch.empty_like(table)
        op = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM
        t_taps = timed(taps_once)
        t_dense = timed(lambda: dist.all_reduce(dummy, op=op))
        tt = torch.tensor([t_taps, t_dense], dtype=torch.float64, device=table.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_taps, t_dense = float(tt[0]), float(tt[1])
        if t_taps - 1.0e-4 > t_dense:
            ok = False
            note = (f"RCCL all-reduce of every gradient, fp32 (timed at start-up: dense table all-reduce {t_dense * 1e3:.3f} ms vs "
                    f"tap exchange + rebuild {t_taps * 1e3:.3f} ms)")
        else:
            timing = f"; timed at start-up: {t_taps * 1e3:.3f} ms vs {t_dense * 1e3:.3f} ms for the dense all-reduce"
    else:
        timing = ""
    model.zero_grad(set_to_none=True)
    model.train(was_training)
    if ok:
        return (f"taps of the word-table gradient all-gathered (fp32, {grad_sync.tap.n * 8 / 1e6:.1f} MB per rank instead of a "
                f"{table.numel() * 4 / 1e6:.0f} MB all-reduce), the other gradients all-reduced; checked against the dense "
                f"all-reduce at start-up (max |diff| {err:.1e}){timing}")
    RF.set_tap_sink(None)
    grad_sync.tap = None
    return note or f"RCCL all-reduce of every gradient, fp32 (tap exchange failed its start-up check: {err})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--comm-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="wire format of the word-table gradient all-reduce (N > 1); fp32 is exact")
    ap.add_argument("--exchange", choices=["auto", "taps", "dense"], default="auto",
                    help="N > 1, word-table gradient: all-gather its taps (4.3 MB per rank, rebuilt on every rank: 0.16 / 0.23 / "
                         "0.37 ms at 2 / 4 / 8 ranks) or all-reduce the dense 60 MB gradient; auto times both on the job's own "
                         "ranks at start-up and keeps the faster one")
    ap.add_argument("--torch-optim", action="store_true",
                    help="clip_grad_norm_ + torch.optim.Adam(fused) instead of the two-launch HipClipAdam")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run --nproc-per-node N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    from review_based_recommender_amd import _lib
    from review_based_recommender_amd.train_step import GraphedTrainStep, make_optimizer, train_step
    _lib.lib()   # fail loudly now if librbr_hip.so is missing

    # rehearsal aids for a one-GPU box (never set by the driver): all ranks on device 0, gloo instead of RCCL
    if os.environ.get("RBR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("RBR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cfg = synth.DEEPCONN_CFGS[WORKLOAD]

    grad_sync = None
    exchange_note = None
    if world > 1:
        import torch.distributed as dist
        from review_based_recommender_amd.distributed import GradAllReduce, init_process_group_from_env
        init_process_group_from_env(backend)

    model = build_model(cfg, device)          # identical parameters on every rank (same seed)
    model.train()
    use_graph = not a.no_graph
    opt = make_optimizer(model, capturable=use_graph, hip_clip_adam=not a.torch_optim)
    args, ratings = batch_on(cfg, 1 + rank, device)   # each rank owns a different shard
    if world > 1:
        use_taps = a.comm_dtype == "fp32" and a.exchange in ("taps", "auto")
        grad_sync = GradAllReduce(model, comm_dtype=torch.bfloat16 if a.comm_dtype == "bf16" else None,
                                  tap_table=model.word_embeddings.embedding.weight if use_taps else None)
        if use_taps:
            exchange_note = verify_tap_exchange(model, args, ratings, grad_sync, dist, choose_by_time=(a.exchange == "auto"))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    launch_note = None
    if use_graph:
        try:
            stepper = GraphedTrainStep(model, opt, args, ratings, grad_sync=grad_sync)
        except Exception as e:      # a capture the runtime refuses must not cost the measurement: launch eagerly instead
            use_graph = False
            launch_note = f"hipGraph capture failed ({type(e).__name__}: {str(e)[:120]}); eager launches"
            torch.cuda.synchronize()
    if use_graph:
        def step():
            stepper()      # the batch is resident in the graph's input buffers (where a loader's H2D copy lands)
    else:
        def step():
            train_step(model, opt, args, ratings, grad_sync=grad_sync)

    for _ in range(a.warmup):
        step()
    barrier()
    if not use_graph:
        _lib.TIMER.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_graph:
        # HIP-event pass over the same K steps, launched eagerly (events cannot be read inside a replayed graph)
        _lib.TIMER.start()
        for _ in range(a.steps):
            train_step(model, opt, args, ratings, grad_sync=grad_sync)
        barrier()
    _lib.TIMER.stop()
    ksum = _lib.TIMER.summary()

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

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        pairs = cfg["B"] * a.steps * world
        flops = conv_fwd_flops(cfg)
        masks = torch.cat([args[2], args[3]])
        dense_mode = "textcnn_conv_fwd" in ksum
        out = {
            "metric": "(user,item) pairs/sec, DeepCoNN train step (fwd+MSE+bwd+clip+Adam), bsz256 2x512tok",
            "value": round(pairs / elapsed, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DeepCoNN cfg2: batch 256 pairs/GPU, 2x512-token docs, D=300, conv widths 3/5/7 x 50, "
                                   "latent 32, V=50002, fp32, Zipf ids", "global_batch": cfg["B"] * world,
                       "parallelism": f"dp{world}", "launch": launch_note or ("hipGraph replay" if use_graph else "eager"),
                       "optimizer": "torch clip_grad_norm_ + fused Adam" if a.torch_optim else "HipClipAdam (clip + Adam, 2 launches)",
                       "grad_exchange": None if world == 1 else (exchange_note or f"RCCL all-reduce, {a.comm_dtype} wire format, before the clip")},
            "fwd_only_pairs_per_s": round(cfg["B"] / fwd_s, 1),
            "kernels_ms": {k: round(v[1], 4) for k, v in ksum.items()},
            "kernel_timing": ("HIP events around the C-ABI launches, eager pass over the same steps after the timed region"
                              if use_graph else "HIP events around the C-ABI launches inside the timed region"),
        }
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
            # default build: the conv runs in its token-product form.  Its contraction -- one launch of the MFMA
            # kernel over the DISTINCT tokens of the batch -- is the longest kernel of the step.  Algorithmic FLOPs of
            # that formulation: 2 * distinct tokens * D * sum(kz * channels)  (DESIGN.md section 4).
            gemm_calls, gemm_ms = ksum["textcnn_prod_table"]
            ids_all = torch.cat([args[0], args[1]])
            n_distinct = int(torch.unique(ids_all[masks]).numel())
            cp = sum(k * (cfg["H"] // len(cfg["kz"])) for k in cfg["kz"])
            gemm_flops = 2.0 * n_distinct * cfg["D"] * cp
            ach = gemm_flops / (gemm_ms * 1e-3) / 1e12
            conv_ms = sum(ksum[k][1] for k in ("textcnn_prod_prepare", "textcnn_prod_table", "textcnn_prod_pool"))
            out["roofline"] = {
                "bound": "mfma", "kernel": "prod_gemm_kernel<60>: T = table[distinct tokens] @ Wprod "
                                           "(v_mfma_f32_32x32x2_f32, rows gathered by LDS-DMA)",
                "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": measured_traffic("r01_prod_table_pmc.json"),
                "traffic_source": "profiles/r01_prod_table_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                "flops_per_launch": gemm_flops, "avg_launch_ms": round(gemm_ms, 4), "launches_timed": gemm_calls,
                "distinct_tokens": n_distinct, "positions": int(masks.numel()),
                "note": "longest kernel of the default (token-product) step; the dense conv it replaces is "
                        f"{flops / 1e9:.1f} GFLOP per launch (RBR_CONV_MODE=dense prices that kernel)"}
            out["conv_stage"] = {
                "formulation": "token-product: T = table[distinct tokens] @ W (f32 MFMA), then gather-add + max-pool",
                "ms": round(conv_ms, 4), "dense_conv_flops_replaced": flops,
                "dense_equivalent_TFLOPs": round(flops / (conv_ms * 1e-3) / 1e12, 1)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
