"""One-process-per-GPU data parallelism for the review-encoder train step.

Replaces the reference's `torch.nn.DataParallel` (trainer/train_deepconn_pp.py:129-131: single
process, per-step parameter broadcast, input scatter, output gather and gradient reduce onto
GPU 0) with the MI355X layout: parameters are replicated once, every rank runs the HIP
forward/backward on its own shard of the batch, and the gradients are summed with ONE exchange
step -- an RCCL all-reduce over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo"
is used by the CPU tests) -- before clip_grad_norm_, so that clipping sees the global gradient
exactly as DataParallel's device-0 gradient does (:165-167).

Equal shards: mean-of-shard-means == the global MSE mean, hence grad = (1/N) * sum_rank grad_rank.

Buckets: the dense parameters (conv banks, heads, id embeddings: 1.2 MB at the
cfg2 shape) are flattened into one latency-bound bucket that is issued first; the word-table
gradient (60 MB fp32) is bandwidth-bound and reduced in place, without a staging copy.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

BIG_BUCKET_ELEMS = 1 << 20   # gradients at least this large are reduced in place, on their own


def init_process_group_from_env(backend: Optional[str] = None) -> None:
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run)."""
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")),
                            world_size=int(os.environ.get("WORLD_SIZE", "1")))


def broadcast_parameters(model: nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank `src`'s parameters (done once, not per step)."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class TapExchange:
    """Exchange of the word-table gradient in tap form (csrc/textcnn_prod.hip, "data-parallel tap exchange").

    With identical conv weights on every rank the table gradient of the max-pooled TextCNN is
    dtable = ((1/N) sum_r G_r) @ Wprod^T, and G_r has one non-zero per (document, channel, tap): the ranks all-gather
    n_docs * C * KF (token, value) pairs (4.3 MB at the cfg2 shape) instead of all-reducing the dense [V, D] gradient
    (60 MB), and each rebuilds the averaged gradient locally.  The rebuild accumulates in 64-bit fixed point with
    integer atomics, so every rank computes bit-identical rows and the replicas stay in lock-step.

    Handles ONE un-gated conv call over the table per step (DeepCoNN++); a second call, or a model whose table also
    receives dense gradient pieces, falls back to the dense all-reduce (GradAllReduce does that when `pending` is False)."""

    def __init__(self, table: nn.Parameter, group=None, owner: bool = False, optimizer=None):
        """owner=True: the OWNER-PARTITIONED rebuild.  Every rank still receives all taps, but turns only the taps of "its" tokens
        (token t belongs to rank t % N) into gradient rows -- sort and row build shrink to 1/N per rank -- and the ranks then
        all-gather their slabs of ceil(V/N) rows: every rank ends up with the same averaged gradient, row t at
        [t % N][t // N].  `optimizer` (a train_step.HipClipAdam) takes it in that layout (functional.RowGradient over a static
        token -> row map: no permuting copy); any other optimizer gets the dense [V, D] tensor."""
        self.table = table
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.owner = bool(owner)
        self.optimizer = optimizer
        self.slabs = self.row_map = self.sq = self.overflow = None
        self.overflow_fallbacks = 0   # owner mode: steps whose gradient was rebuilt on every rank because an owner's share overflowed
        self._test_force_overflow = False
        self.n = 0
        self.tok = self.val = self.dtable = self.ws = None
        self.desc = self.weights = None
        self.calls = 0
        self.captured = None    # (desc, weights) of a backward that was recorded into a hipGraph
        self.dense_fallbacks = 0   # conv calls over this table that went the dense all-reduce way (second call of a step,
                                   # unsupported shape): the advertised exchange was not used for them

    # ---- called from functional._TextCNN.backward
    def accepts(self, table: torch.Tensor, desc, L_) -> bool:
        import ctypes as C
        if table.data_ptr() != self.table.data_ptr():
            return False             # a foreign table (functional routes by table, so this is a stale registration)
        if self.calls > 0:
            self.calls += 1          # a second conv over the table in this step: this one goes the dense way
            self._note_fallback("a second conv over the word table in one step")
            return False
        ok = L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(desc), self.world) > 0
        if not ok:
            self._note_fallback("the conv shape is outside what the tap rebuild supports")
        return ok

    def _note_fallback(self, why: str) -> None:
        self.dense_fallbacks += 1
        if self.dense_fallbacks == 1 and self.rank == 0:
            print(f"[TapExchange] dense all-reduce of the word-table gradient instead of the tap exchange: {why}", flush=True)

    def close(self) -> None:
        """Uninstalls the sink (the table's backward produces its dense gradient again)."""
        from . import functional as RF
        RF.set_tap_sink(None, self.table)

    def local_buffers(self, n: int, dev):
        if self.tok is None or self.n != n or self.tok.device != dev:
            self.n = n
            self.tok = torch.empty(self.world * n, dtype=torch.int32, device=dev)
            self.val = torch.empty(self.world * n, dtype=torch.float32, device=dev)
        lo = self.rank * n
        return self.tok[lo:lo + n], self.val[lo:lo + n]

    def record(self, desc, weights) -> None:
        self.desc, self.weights = desc, weights
        self.calls += 1
        if torch.cuda.is_current_stream_capturing():
            self.captured = (desc, weights)      # no Python runs when the graph is replayed: see mark_replayed()

    def mark_replayed(self) -> None:
        """The recorded backward has just been replayed (GraphedTrainStep): its taps are in the buffers."""
        if self.captured is not None:
            self.desc, self.weights = self.captured
            self.calls = 1

    # ---- called from GradAllReduce
    @property
    def pending(self) -> bool:
        return self.desc is not None

    def start(self):
        """Issues the all-gather of the taps; returns the work handles."""
        n, lo = self.n, self.rank * self.n
        if dist.get_backend(self.group) == "nccl":
            return [dist.all_gather_into_tensor(self.tok, self.tok[lo:lo + n], group=self.group, async_op=True),
                    dist.all_gather_into_tensor(self.val, self.val[lo:lo + n], group=self.group, async_op=True)]
        hs = []
        for full in (self.tok, self.val):          # gloo (rehearsals / CPU-hosted tests): list form, out-of-place input
            hs.append(dist.all_gather(list(full.split(n)), full[lo:lo + n].clone(), group=self.group, async_op=True))
        return hs

    def finish(self) -> None:
        """Rebuilds the averaged table gradient from all ranks' taps into table.grad; a step that also sent part of
        the table gradient the dense way (second conv call) keeps that part in table.grad for the dense all-reduce."""
        import ctypes as C
        from . import _lib
        L_ = _lib.lib()
        dev = self.tok.device
        if self.owner:
            return self._finish_owner(L_, dev)
        if self.dtable is None:
            self.dtable = torch.empty_like(self.table)
            self.ws = torch.empty(L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(self.desc), self.world), dtype=torch.uint8,
                                  device=dev)
        _lib.check(L_.rbr_textcnn_dtable_from_taps(C.byref(self.desc), self.world, _lib.dev_ptr(self.tok, torch.int32, "tap tokens"),
                                                   _lib.dev_ptr(self.val, torch.float32, "tap values"),
                                                   _lib.ptr_array(self.weights, torch.float32, "conv weight"),
                                                   self.ws.data_ptr(), _lib.dev_ptr(self.dtable, torch.float32, "dtable"),
                                                   _lib.current_stream()), "rbr_textcnn_dtable_from_taps")
        dense_part = self.table.grad if self.calls > 1 else None      # already averaged by the dense all-reduce
        if dense_part is not None and dense_part is not self.dtable:
            self.dtable.add_(dense_part)
        self.table.grad = self.dtable
        self.desc = self.weights = None
        self.calls = 0


    def _finish_owner(self, L_, dev) -> None:
        import ctypes as C
        from . import _lib, functional as RF
        V, D = int(self.table.shape[0]), int(self.table.shape[1])
        if self.slabs is None:
            v_own = int(L_.rbr_textcnn_taps_owner_rows(C.byref(self.desc), self.world))
            self.v_own = v_own
            # [rank][row], one extra row per rank whose first element carries the sum of squares of the rank's rows (the clip's
            # norm needs nothing else from the other ranks); the all-gather fills every rank's part
            self.slabs = torch.zeros(self.world, v_own + 1, D, dtype=torch.float32, device=dev)
            t = torch.arange(V, device=dev)
            self.row_map = ((t % self.world) * (v_own + 1) + t // self.world).to(torch.int32)
            self.sq = torch.zeros(self.world, dtype=torch.float32, device=dev)
            self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
            self.ws = torch.empty(L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(self.desc), self.world), dtype=torch.uint8,
                                  device=dev)
        v_own = self.v_own
        own = self.slabs[self.rank]
        self.overflow.zero_()                      # the flag is sticky inside the library: per step here
        _lib.check(L_.rbr_textcnn_dtable_from_taps_owner(C.byref(self.desc), self.world, self.rank,
                                                         _lib.dev_ptr(self.tok, torch.int32, "tap tokens"),
                                                         _lib.dev_ptr(self.val, torch.float32, "tap values"),
                                                         _lib.ptr_array(self.weights, torch.float32, "conv weight"),
                                                         self.ws.data_ptr(), _lib.dev_ptr(own, torch.float32, "slab"),
                                                         _lib.dev_ptr(self.overflow, torch.int32, "overflow"),
                                                         _lib.current_stream()), "rbr_textcnn_dtable_from_taps_owner")
        if self._test_force_overflow:              # tests only: exercise the fallback protocol at world sizes that cannot overflow
            self.overflow.fill_(1)
        rows = own[:v_own]
        own[v_own, 0] = torch.linalg.vector_norm(rows).square()
        own[v_own, 1] = self.overflow[0].to(torch.float32)      # the rank's overflow flag travels with its slab
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(self.slabs.view(-1), own.reshape(-1), group=self.group)      # in place: own is its slice
        else:
            dist.all_gather(list(self.slabs.view(self.world, -1).unbind(0)), own.reshape(-1).clone(), group=self.group)
        dense_part = self.table.grad if self.calls > 1 else None      # a second conv's part, already averaged densely
        # A rank that owned more taps than its sort is sized for (twice the even share: the Zipf head on one owner) dropped the
        # surplus: its slab is incomplete.  Every rank reads the SAME gathered flags, so all of them take the same branch -- no
        # rank is left waiting at a collective -- and rebuild this step's gradient the replicated way from the taps they already
        # hold (bit-identical on every rank).  The read is a host synchronisation per step: the price of the owner mode's
        # smaller sort, paid only in that mode.
        if bool((self.slabs[:, v_own, 1] != 0).any().item()):
            self.overflow_fallbacks += 1
            if self.overflow_fallbacks == 1 and self.rank == 0:
                print("[TapExchange] owner rebuild: a rank owned more taps than twice the even share; this step (and any like it) "
                      "rebuilds the word-table gradient on every rank instead", flush=True)
            if self.dtable is None:
                self.dtable = torch.empty_like(self.table)
            _lib.check(L_.rbr_textcnn_dtable_from_taps(C.byref(self.desc), self.world, _lib.dev_ptr(self.tok, torch.int32, "tap tokens"),
                                                       _lib.dev_ptr(self.val, torch.float32, "tap values"),
                                                       _lib.ptr_array(self.weights, torch.float32, "conv weight"),
                                                       self.ws.data_ptr(), _lib.dev_ptr(self.dtable, torch.float32, "dtable"),
                                                       _lib.current_stream()), "rbr_textcnn_dtable_from_taps")
            if dense_part is not None and dense_part is not self.dtable:
                self.dtable.add_(dense_part)
            if self.optimizer is not None and hasattr(self.optimizer, "_row_grads"):
                self.optimizer._row_grads.pop(self.table, None)       # a previous step's exchanged rows must not be read again
            self.table.grad = self.dtable
            self.desc = self.weights = None
            self.calls = 0
            return
        self.sq.copy_(self.slabs[:, v_own, 0])
        rg = RF.RowGradient(self.table, self.slabs.view(-1, D), self.sq, self.row_map.data_ptr(), (self.row_map, self.slabs))
        taken = False
        if dense_part is None and self.optimizer is not None and hasattr(self.optimizer, "put_exchanged_rows"):
            taken = self.optimizer.put_exchanged_rows(self.table, rg)
        if not taken:
            dense = rg.to_dense()
            self.table.grad = dense if dense_part is None else dense.add_(dense_part)
        self.desc = self.weights = None
        self.calls = 0

    def check(self) -> None:
        """Owner mode: kept for callers of earlier rounds.  An overflowing step no longer leaves a wrong gradient behind --
        _finish_owner detects it on every rank from the gathered flags and rebuilds the replicated way -- so there is nothing
        left to raise; `overflow_fallbacks` counts such steps."""
        return None


class GradAllReduce:
    """Callable `grad_sync(model)` hook for train_step(): averages .grad over the process group."""

    def __init__(self, model: nn.Module, group=None, comm_dtype: Optional[torch.dtype] = None,
                 tap_table: Optional[nn.Parameter] = None, owner: bool = False, optimizer=None):
        """`tap_table`: the word-table parameter whose gradient is exchanged in tap form (TapExchange) instead of being
        all-reduced densely; None keeps the dense all-reduce for every parameter.  `owner` / `optimizer`: the
        owner-partitioned rebuild of TapExchange and the optimizer that takes its row-form result."""
        self.tap = None
        if tap_table is not None and dist.get_world_size(group) > 1:
            from . import functional as RF
            self.tap = TapExchange(tap_table, group, owner=owner, optimizer=optimizer)
            RF.set_tap_sink(self.tap)          # keyed by the table: other models' convs never reach this sink
        self.group = group
        self.world = dist.get_world_size(group)
        self.comm_dtype = comm_dtype   # e.g. torch.bfloat16 halves the table bucket; None = exact fp32
        params = [p for p in model.parameters() if p.requires_grad]
        if self.world > 1:
            # the exchange reads every gradient as a dense .grad: an optimizer's compact row-gradient hand-off
            # (train_step.HipClipAdam) is switched off for this model's tables
            from . import functional as RF
            for p in params:
                RF.set_row_grad_sink(p, None)
        self.big: List[nn.Parameter] = [p for p in params if p.numel() >= BIG_BUCKET_ELEMS]
        self.small: List[nn.Parameter] = [p for p in params if p.numel() < BIG_BUCKET_ELEMS]

    def __call__(self, model: nn.Module = None) -> None:
        if self.world == 1:
            return
        # RCCL/NCCL averages in the collective itself; gloo (CPU tests) sums and the mean is taken afterwards
        native_avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if native_avg else dist.ReduceOp.SUM
        inv = 1.0 / self.world
        handles = []
        tap_pending = self.tap is not None and self.tap.pending
        if tap_pending:
            if self.tap.calls == 1:
                self.tap.table.grad = None          # a stale gradient of the previous step is not part of this one
            handles += self.tap.start()
        elif self.tap is not None:
            self.tap.calls = 0
        small = [p for p in self.small if p.grad is not None]
        flat = None
        if small:
            # one cat kernel in, one multi-tensor copy kernel out (not 2 x 17 small copies per step)
            flat = torch.cat([p.grad.reshape(-1) for p in small])
            handles.append(dist.all_reduce(flat, op=op, group=self.group, async_op=True))
        staged = []
        for p in self.big:
            if p.grad is None:
                continue
            if self.comm_dtype is not None and self.comm_dtype != p.grad.dtype:
                buf = p.grad.to(self.comm_dtype)
                staged.append((p, buf))
                handles.append(dist.all_reduce(buf, op=op, group=self.group, async_op=True))
            else:
                handles.append(dist.all_reduce(p.grad, op=op, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if flat is not None:
            if not native_avg:
                flat.mul_(inv)
            views = [v.view_as(p.grad) for v, p in zip(flat.split([p.grad.numel() for p in small]), small)]
            torch._foreach_copy_([p.grad for p in small], views)
        for p, buf in staged:
            p.grad.copy_(buf)
        if not native_avg:
            for p in self.big:
                if p.grad is not None:
                    p.grad.mul_(inv)
        if tap_pending:
            self.tap.finish()


    def close(self) -> None:
        """Removes the tap sink this hook installed (no-op for the dense exchange)."""
        if self.tap is not None:
            self.tap.close()
            self.tap = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def shard_batch(tensors, rank: int, world: int):
    """Equal contiguous dim-0 shards of a global batch (DataParallel's scatter, :130)."""
    out = []
    for t in tensors:
        n = t.shape[0]
        assert n % world == 0, "global batch must divide evenly across ranks"
        s = n // world
        out.append(t[rank * s:(rank + 1) * s].contiguous())
    return tuple(out)
