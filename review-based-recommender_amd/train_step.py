"""The trainer's optimisation step (trainer/train_deepconn_pp.py:161-168, train_narre.py:163-169,
train_dual_att.py:157-163), reproduced above the module boundary so bench.py and the parity
tests drive the HIP modules exactly as the reference trainers drive theirs:

    optimizer.zero_grad(); y = model(*batch); loss = MSELoss()(y, ratings); loss.backward()
    [data-parallel: all-reduce the gradients]
    clip_grad_norm_(model.parameters(), max_grad_norm); optimizer.step()

Nothing here calls .item(): the step stays asynchronous on the HIP stream (the reference's two
loss.item() calls per step, :171-172, are logging, not part of the math).
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as RF

LR = 2e-3            # default_deepconn_pp.json:24
MAX_GRAD_NORM = 5.0  # default_deepconn_pp.json:27


def make_optimizer(model: nn.Module, lr: float = LR, fused: bool | None = None,
                   capturable: bool = False, hip_clip_adam: bool = False) -> torch.optim.Optimizer:
    """torch.optim.Adam(model.parameters(), lr=args.lr)  (train_deepconn_pp.py:135).

    `fused=None` picks torch's single-kernel ("fused") Adam implementation when every parameter lives on
    a HIP device and the default multi-kernel one otherwise; both compute the same update.
    `capturable=True` keeps the step counter on the device so the update can be recorded into a hipGraph
    (GraphedTrainStep).  `hip_clip_adam=True` returns HipClipAdam instead: same update, with the trainer's
    clip_grad_norm_ folded into the optimizer pass (train_step() then calls its clip_and_step)."""
    params = list(model.parameters())
    if hip_clip_adam:      # clip + Adam as two HIP launches (HipClipAdam below); always graph-capturable
        return HipClipAdam(params, lr=lr)
    if fused is None:
        fused = len(params) > 0 and all(p.is_cuda for p in params)
    if fused:
        return torch.optim.Adam(params, lr=lr, fused=True, capturable=capturable)
    return torch.optim.Adam(params, lr=lr, capturable=capturable)


class HipClipAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr)  +  the clip_grad_norm_ in front of its step  (train_deepconn_pp.py:135,166-167)
    as two HIP launches over all parameter tensors (csrc/clip_adam.hip): torch's sequence is ~10 kernels and three
    full passes over the gradients, dominated by the 60 MB word table.

    Same update rule and state names as torch.optim.Adam (amsgrad / weight_decay / maximize unsupported, as the
    reference trainers never set them); `state_dict()` interchanges with it.  The step counter lives on the device,
    so `clip_and_step` can be recorded into a hipGraph.  Parameters must be contiguous fp32 HIP tensors."""

    MAX_TENSORS = 64      # RBR_OPT_MAX_TENSORS

    ROW_GRAD_MIN_ROWS = 4096      # tables at least this tall take their gradient in compact row form

    def __init__(self, params, lr: float = LR, betas=(0.9, 0.999), eps: float = 1e-8, row_grads: bool = True):
        """row_grads: embedding tables among the parameters (2-D, >= ROW_GRAD_MIN_ROWS rows, row length % 4 == 0) receive the
        gradient of a token-product conv as the rows of the batch's tokens (functional.RowGradient) instead of a dense
        [V, D] tensor whose other rows are zeros -- but only for a backward that runs inside this optimizer's row_grad_scope()
        (train_step() and GraphedTrainStep open it around the forward + backward of a step THIS optimizer will finish): `.grad`
        of such a table then stays None after backward -- call materialize_grads() to see the dense gradient -- and the clip +
        Adam launches neither read nor re-write the zero rows.  Same parameters and state as with the dense gradient, bit for
        bit.  Outside the scope (a hand-written loop, another optimizer on the same model, an external clip_grad_norm_) every
        backward leaves the ordinary dense .grad."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._ws = None
        self._gnorm = None
        self._row_grads = {}          # parameter -> functional.RowGradient of the last backward
        self._row_tables = []
        self._row_scope = 0           # > 0: inside row_grad_scope() -- the only time wants_row_grad() says yes
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            row_grads = False         # a data-parallel exchange reads every gradient as a dense .grad
        if os.environ.get("RBR_ROW_GRADS", "1") == "0":
            row_grads = False
        if row_grads:
            for g in self.param_groups:
                for p in g["params"]:
                    if p.is_cuda and p.dim() == 2 and p.shape[0] >= self.ROW_GRAD_MIN_ROWS and p.shape[1] % 4 == 0 \
                            and p.dtype == torch.float32 and p.numel() < (1 << 32):
                        RF.set_row_grad_sink(p, self)
                        self._row_tables.append(p)

    # ---- functional.set_row_grad_sink protocol
    def wants_row_grad(self, table) -> bool:
        return self._row_scope > 0 and any(table.data_ptr() == p.data_ptr() for p in self._row_tables)

    @contextlib.contextmanager
    def row_grad_scope(self):
        """with opt.row_grad_scope(): forward; loss.backward() -- the caller's promise that THIS optimizer's clip_and_step() /
        step() is what consumes the gradients of that backward: its tables then receive their gradient in compact row form.
        (The registry in functional holds the optimizer weakly and asks wants_row_grad() per backward, so an optimizer that
        merely exists -- beside torch's Adam on the same model, or after its last step -- changes nothing.)"""
        self._row_scope += 1
        try:
            yield self
        finally:
            self._row_scope -= 1

    def put_row_grad(self, table, rg) -> None:
        p = next(q for q in self._row_tables if q.data_ptr() == table.data_ptr())
        old = self._row_grads.pop(p, None)
        if old is not None:               # a second backward before the step: accumulate the ordinary way
            dense = old.to_dense()
            p.grad = dense if p.grad is None else p.grad + dense
        self._row_grads[p] = rg

    def put_exchanged_rows(self, table, rg) -> bool:
        """A data-parallel exchange (distributed.TapExchange, owner mode) hands over the AVERAGED gradient of `table` in row form:
        the next clip_and_step reads it through rg's token -> row map.  False: not a table this optimizer can take that way."""
        p = next((q for g in self.param_groups for q in g["params"] if q.data_ptr() == table.data_ptr()), None)
        if p is None or p.dim() != 2 or p.shape[1] % 4 != 0 or p.dtype != torch.float32 or not p.is_cuda \
                or os.environ.get("RBR_ROW_GRADS", "1") == "0":
            return False
        self._row_grads.clear()           # one row-form table per launch pair
        self._row_grads[p] = rg
        p.grad = None
        return True

    def close(self) -> None:
        """Unregisters the row-gradient hand-off (the tables get dense gradients again)."""
        for p in self._row_tables:
            RF.set_row_grad_sink(p, None)
        self._row_tables = []
        self.materialize_grads()

    @torch.no_grad()
    def materialize_grads(self) -> None:
        """Turns every pending compact row gradient into the dense `.grad` nn.Embedding's backward would have left (after a
        clipping step: the clipped gradient, as clip_grad_norm_ leaves it)."""
        for p, rg in list(self._row_grads.items()):
            dense = rg.to_dense()
            p.grad = dense if p.grad is None else p.grad + dense
        self._row_grads.clear()

    def zero_grad(self, set_to_none: bool = True):
        self._row_grads.clear()
        super().zero_grad(set_to_none=set_to_none)

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def clip_and_step(self, max_grad_norm: float | None = None) -> torch.Tensor:
        """Clips the gradients of ALL parameter groups to a total 2-norm of `max_grad_norm` (None: no clipping),
        applies the Adam update, leaves the clipped gradients in .grad and returns the norm before clipping."""
        import ctypes as C
        from . import _lib
        L_ = _lib.lib()
        for p in [q for q in self._row_grads if q.grad is not None]:      # mixed: a dense gradient exists too -> dense path
            p.grad = p.grad + self._row_grads.pop(p).to_dense()
        todo = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None or p in self._row_grads]
        if not todo:
            raise RuntimeError("HipClipAdam.clip_and_step: no gradients")
        dev = todo[0][1].device
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(L_.rbr_clip_adam_ws_floats(), dtype=torch.float32, device=dev)
            self._gnorm = torch.zeros((), dtype=torch.float32, device=dev)
        st = _lib.current_stream()
        # one launch pair per parameter group and per 64 tensors; the common case (17 tensors, one group) is one pair.
        # With several pairs the norm must still be global: a first sweep with lr = 0 would be wasteful, so more than
        # one pair is only allowed without clipping.
        batches = []
        for g in self.param_groups:
            ps = [p for gg, p in todo if gg is g]
            for k in range(0, len(ps), self.MAX_TENSORS):
                batches.append((g, ps[k:k + self.MAX_TENSORS]))
        if self._row_grads and (len(batches) > 1 or len(self._row_grads) > 1):
            self.materialize_grads()          # one compact table per launch pair: anything else takes the dense path
        if len(batches) > 1 and max_grad_norm is not None:
            # the fused clip needs every gradient in ONE launch pair; beyond 64 tensors (or with several parameter groups) the
            # norm and the in-place scaling are torch's clip_grad_norm_ (same maths, device-side, capturable) and the launch
            # pairs below run without clipping
            gn = nn.utils.clip_grad_norm_([p for _, p in todo], max_grad_norm)
            self._gnorm.copy_(gn)
            max_grad_norm = None
        for g, ps in batches:
            states = [self._state_of(p) for p in ps]
            step = states[0]["step"]
            for s_ in states[1:]:          # the tensors of a batch step together: one shared device counter
                if s_["step"] is not step:
                    s_["step"] = step
            grads = [p.grad for p in ps]          # None for the table whose gradient is in row form
            for p, gr in zip(ps, grads):
                if not (p.is_contiguous() and (gr is None or gr.is_contiguous())):
                    raise RuntimeError("HipClipAdam needs contiguous parameters and gradients")
            numel = (C.c_int64 * len(ps))(*[p.numel() for p in ps])
            b1, b2 = g["betas"]
            ev = _lib.TIMER.record("clip_adam_step")
            args = (len(ps), _lib.ptr_array(ps, torch.float32, "param"), _lib.ptr_array(grads, torch.float32, "grad"),
                    _lib.ptr_array([s_["exp_avg"] for s_ in states], torch.float32, "exp_avg"),
                    _lib.ptr_array([s_["exp_avg_sq"] for s_ in states], torch.float32, "exp_avg_sq"),
                    numel, float(max_grad_norm) if max_grad_norm is not None else 0.0,
                    float(g["lr"]), float(b1), float(b2), float(g["eps"]), _lib.dev_ptr(step, torch.float32, "step"),
                    _lib.dev_ptr(self._gnorm, torch.float32, "gnorm") if len(batches) == 1 else None,
                    _lib.dev_ptr(self._ws, torch.float32, "ws"))
            rowp = [k for k, p in enumerate(ps) if p in self._row_grads]
            if rowp:
                rg = self._row_grads[ps[rowp[0]]]
                crg = _lib.RowGrad(rowp[0], rg.V, rg.D, rg.row_of_token_ptr, rg.rows.data_ptr(), rg.sq.data_ptr(), rg.sq.numel())
                _lib.check(L_.rbr_clip_adam_step_rows(*args, C.byref(crg), st), "rbr_clip_adam_step_rows")
            else:
                _lib.check(L_.rbr_clip_adam_step(*args, st), "rbr_clip_adam_step")
            if ev is not None:
                ev.record()
        return self._gnorm

    @torch.no_grad()
    def step(self, closure=None):
        """Plain Adam step (no clipping), for callers that clip separately: `loss.backward(); clip_grad_norm_(...); opt.step()`
        sees every gradient as a dense .grad, because a backward outside row_grad_scope() never produces the row form.  Should
        rows be pending all the same (the caller opened the scope and then clipped by hand), they are made dense here first --
        late for that clip, which is why the scope is train_step()'s business and not a default."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.clip_and_step(None)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():      # torch.optim.Adam checkpoints hold a CPU float / int step
            if "step" in st and (not torch.is_tensor(st["step"]) or not st["step"].is_cuda):
                dev = st["exp_avg"].device
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=dev)


def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, batch, ratings: torch.Tensor,
               max_grad_norm: float = MAX_GRAD_NORM, grad_sync=None):
    """One step.  `batch` is the tuple passed to model(*batch); models returning a tuple
    (NARRE: pred, u_att, i_att) contribute their first element.  `grad_sync(model)` is the
    data-parallel gradient all-reduce hook (None on one GPU).  Returns (loss, gnorm, pred) tensors."""
    optimizer.zero_grad()
    pred, loss = _forward_loss_backward(model, batch, ratings, optimizer if grad_sync is None else None)
    if grad_sync is not None:
        if isinstance(optimizer, HipClipAdam):
            optimizer.materialize_grads()        # the all-reduce wants every gradient as a dense .grad
        grad_sync(model)
    gnorm = clip_and_step(model, optimizer, max_grad_norm)
    return loss.detach(), gnorm, pred.detach()


def _forward_loss_backward(model: nn.Module, batch, ratings: torch.Tensor, optimizer=None):
    """y = model(*batch); loss = MSELoss()(y, ratings); loss.backward()  (train_deepconn_pp.py:162-165).  The target is
    announced to the forward (functional.fused_loss): a model whose last launch can compute the loss takes it along.
    `optimizer`: the optimizer whose step follows this backward; a HipClipAdam then takes its tables' gradients in compact row
    form (its row_grad_scope() is open for exactly this forward + backward)."""
    scope = optimizer.row_grad_scope() if isinstance(optimizer, HipClipAdam) else contextlib.nullcontext()
    fusable = ratings.is_cuda and ratings.dtype == torch.float32
    with scope:
        with (RF.fused_loss(ratings) if fusable else contextlib.nullcontext()) as req:
            out = model(*batch)
            pred = out[0] if isinstance(out, tuple) else out
            loss = req.loss_for(pred) if req is not None else None
        if loss is not None:
            loss.backward(RF.unit_scalar(pred.device))
            return pred, loss
        return pred, _loss_and_backward(pred, ratings)


def _loss_and_backward(pred: torch.Tensor, ratings: torch.Tensor) -> torch.Tensor:
    """loss = MSELoss()(pred, ratings); loss.backward()  (train_deepconn_pp.py:164-165)."""
    if pred.is_cuda and pred.dtype == torch.float32 and ratings.dtype == torch.float32 and pred.shape == ratings.shape:
        # one launch for the loss AND its gradient; the root gradient is a cached device scalar (no fill per step)
        loss = RF.mse_loss(pred, ratings)
        loss.backward(RF.unit_scalar(pred.device))
    else:
        loss = F.mse_loss(pred, ratings)
        loss.backward()
    return loss


def clip_and_step(model: nn.Module, optimizer: torch.optim.Optimizer, max_grad_norm: float) -> torch.Tensor:
    """clip_grad_norm_(model.parameters(), max_grad_norm); optimizer.step()  (train_deepconn_pp.py:166-167).
    Returns the total gradient norm before clipping."""
    if isinstance(optimizer, HipClipAdam):
        return optimizer.clip_and_step(max_grad_norm)
    gnorm = nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    return gnorm


def _flat_layout(tensors, align: int = 256):
    """Byte offsets of `tensors` laid out one after another in one block; neighbours (2k, 2k+1) of equal shape and dtype
    are packed back to back (no padding between them) so functional.stack_rows() sees one [2n, ...] tensor.  Last entry:
    total bytes."""
    offs, o, k = [], 0, 0
    while k < len(tensors):
        a = tensors[k]
        b = tensors[k + 1] if k + 1 < len(tensors) else None
        o = (o + align - 1) // align * align
        offs.append(o)
        o += a.numel() * a.element_size()
        if b is not None and a.shape == b.shape and a.dtype == b.dtype and a.dim() >= 1:
            offs.append(o)
            o += b.numel() * b.element_size()
            k += 2
        else:
            k += 1
    offs.append((o + align - 1) // align * align)
    return offs


def _flat_views(flat: torch.Tensor, layout, like):
    return [flat[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(layout, like)]


def _lib_mod():
    from . import _lib
    return _lib


def _capture_stream(dev) -> torch.cuda.Stream:
    """A stream to record a graph on, with the library's per-stream state (functional.prepare_stream_state) already in place."""
    cs = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cs):
        RF.prepare_stream_state(dev)
    torch.cuda.synchronize(dev)
    return cs


class GraphedForward:
    """model(*batch) under torch.no_grad(), recorded once into a hipGraph and replayed: the eval / serving forward of a
    fixed batch shape (the trainers' validation loops, trainer/train_deepconn_pp.py:171-196: ~25 short kernels whose eager
    launches leave the GPU waiting on Python).  The model's train/eval mode at construction is what the graph holds -- put
    the model in eval() first.  __call__ copies a batch of the same shapes into the static input block (one copy when it was
    pack()ed) and replays; the returned prediction tensor is static: overwritten by the next replay."""

    def __init__(self, model: nn.Module, batch, warmup: int = 2, capture_error_mode: str = "global"):
        batch = list(batch)
        if not batch or not batch[0].is_cuda:
            raise RuntimeError("GraphedForward needs HIP tensors")
        self.model = model
        self._layout = _flat_layout(batch)
        self._flat = torch.empty(self._layout[-1], dtype=torch.uint8, device=batch[0].device)
        views = _flat_views(self._flat, self._layout, batch)
        for v, src in zip(views, batch):
            v.copy_(src)
        self.batch = tuple(views)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():      # warm-up off the default stream (allocator, lazy module state)
            for _ in range(warmup):
                model(*self.batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._capture_stream = _capture_stream(batch[0].device)
        with torch.no_grad(), _lib_mod().capture_guard(self._capture_stream), \
                torch.cuda.graph(self.graph, stream=self._capture_stream, capture_error_mode=capture_error_mode):
            out = model(*self.batch)
        self.pred = out[0] if isinstance(out, tuple) else out

    def matches(self, batch) -> bool:
        """True when `batch` has the shapes and dtypes the graph was recorded with (a ragged last batch does not)."""
        return len(batch) == len(self.batch) and all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(batch, self.batch))

    def pack(self, batch) -> torch.Tensor:
        blob = torch.empty_like(self._flat)
        for v, src in zip(_flat_views(blob, self._layout, list(batch)), batch):
            v.copy_(src)
        return blob

    def __call__(self, batch=None, packed: torch.Tensor | None = None) -> torch.Tensor:
        if packed is not None:
            if packed.shape != self._flat.shape or packed.dtype != torch.uint8:
                raise RuntimeError("packed batch does not match the forward's input layout (use GraphedForward.pack)")
            self._flat.copy_(packed, non_blocking=True)
        if batch is not None:
            if not self.matches(batch):
                raise RuntimeError("batch shapes differ from the recorded forward's (run ragged batches eagerly)")
            for dst, src in zip(self.batch, batch):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.pred


def _count_kernel_nodes(graph: torch.cuda.CUDAGraph) -> int:
    """hipGraphGetNodes + hipGraphNodeGetType over the captured hipGraph_t (kernel nodes only: type 0)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    g = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    if hip.hipGraphGetNodes(g, None, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (C.c_void_p * max(1, n.value))()
    if hip.hipGraphGetNodes(g, nodes, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    kernels = 0
    for k in range(n.value):
        ty = C.c_int(-1)
        if hip.hipGraphNodeGetType(C.c_void_p(nodes[k]), C.byref(ty)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        kernels += int(ty.value == 0)          # hipGraphNodeTypeKernel
    return kernels


class _StepSlot:
    """One input block of a GraphedTrainStep and the graph(s) recorded over it."""
    __slots__ = ("flat", "batch", "ratings", "g_fwd_bwd", "g_update", "static_grads", "loss", "gnorm", "pred")


class GraphedTrainStep:
    """train_step() recorded once into a hipGraph and replayed: the step is ~70 short kernels (0.8 ms of GPU
    work at the cfg2 shape), so launching them one by one from Python leaves the GPU waiting on the host.

    Everything data-dependent in the HIP path (distinct-token list, work lists, argmax windows) lives in device
    memory and every launch has a shape-only grid, so a recorded step is valid for any batch of the same shape:
    __call__ copies the new batch into the static input buffers and replays.  With a data-parallel
    `grad_sync` the step is recorded as two graphs (zero_grad+forward+backward | clip+Adam) sharing one memory
    pool, and the RCCL all-reduce runs eagerly between them.

    `slots` > 1 records the step once per INPUT SLOT: each slot has its own input block, so a loader stages batch
    i + 1 into slot (i + 1) % slots (one host-to-device copy, `stage` / `slot_inputs`) while slot i % slots is replayed,
    and the step itself starts with no device-to-device copy of its inputs (`__call__(slot=k)`).  Parameters,
    optimizer state and the optimizer's own buffers are shared by the slots; every slot keeps its own graph memory.

    The optimizer must have been built with make_optimizer(..., capturable=True)."""

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, batch, ratings: torch.Tensor,
                 max_grad_norm: float = MAX_GRAD_NORM, grad_sync=None, warmup: int = 3, capture_error_mode: str | None = None,
                 slots: int = 1, keep_graph: bool = False):
        if not ratings.is_cuda:
            raise RuntimeError("GraphedTrainStep needs HIP tensors")
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self.model, self.optimizer, self.grad_sync, self.max_grad_norm = model, optimizer, grad_sync, max_grad_norm
        self._keep_graph = bool(keep_graph)      # keeps the hipGraph_t behind the executable: kernel_launches() can count its nodes
        # every input of the step lives in ONE block (`flat`): a loader hands over a batch with a single device-to-device
        # (or host-to-device) copy, and the two towers' inputs are neighbours, so the models stack them as a view
        self._layout = _flat_layout(list(batch) + [ratings])
        self._slots = []
        for _ in range(slots):
            sl = _StepSlot()
            sl.flat = torch.empty(self._layout[-1], dtype=torch.uint8, device=ratings.device)
            views = _flat_views(sl.flat, self._layout, list(batch) + [ratings])
            for v, src in zip(views, list(batch) + [ratings]):
                v.copy_(src)
            sl.batch, sl.ratings = tuple(views[:-1]), views[-1]
            sl.g_update = sl.static_grads = None
            self._slots.append(sl)
        first = self._slots[0]
        # the warm-up steps below must not count as training: parameters and Adam state are put back in place
        # (same storage -- the recorded graph keeps their addresses) once the graph exists
        saved_params = [p.detach().clone() for p in model.parameters()]
        saved_state = {p: {k: v.clone() for k, v in optimizer.state[p].items() if torch.is_tensor(v)}
                       for group in optimizer.param_groups for p in group["params"] if p in optimizer.state}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up off the default stream: allocator, lazy optimizer state, occupancy queries
            for _ in range(warmup):
                train_step(model, optimizer, first.batch, first.ratings, max_grad_norm, grad_sync)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # with a process group alive, RCCL's watchdog thread polls events while we record: only this thread's calls
        # may be policed by the capture ("thread_local"), as torch recommends for captures next to NCCL work
        mode = capture_error_mode or ("thread_local" if grad_sync is not None else "global")
        self._capture_stream = _capture_stream(ratings.device)
        cs = self._capture_stream
        for sl in self._slots:
            optimizer.zero_grad(set_to_none=True)
            sl.g_fwd_bwd = torch.cuda.CUDAGraph(keep_graph=True) if self._keep_graph else torch.cuda.CUDAGraph()
            if grad_sync is None:
                with _lib_mod().capture_guard(cs), torch.cuda.graph(sl.g_fwd_bwd, stream=cs, capture_error_mode=mode):
                    sl.loss, sl.gnorm, sl.pred = train_step(model, optimizer, sl.batch, sl.ratings, max_grad_norm)
                sl.static_grads = [(p, p.grad) for p in model.parameters()]
            else:
                with _lib_mod().capture_guard(cs), torch.cuda.graph(sl.g_fwd_bwd, stream=cs, capture_error_mode=mode):
                    optimizer.zero_grad()
                    pred, loss = _forward_loss_backward(model, sl.batch, sl.ratings)
                    sl.loss, sl.pred = loss.detach(), pred.detach()
                    if isinstance(optimizer, HipClipAdam):
                        optimizer.materialize_grads()
                grad_sync(model)
                sl.static_grads = [(p, p.grad) for p in model.parameters()]
                sl.g_update = torch.cuda.CUDAGraph(keep_graph=True) if self._keep_graph else torch.cuda.CUDAGraph()
                with _lib_mod().capture_guard(cs), torch.cuda.graph(sl.g_update, pool=sl.g_fwd_bwd.pool(), stream=cs, capture_error_mode=mode):
                    sl.gnorm = clip_and_step(model, optimizer, max_grad_norm)
        with torch.no_grad():
            for p, v in zip(model.parameters(), saved_params):
                p.copy_(v)
            for group in optimizer.param_groups:
                for p in group["params"]:
                    for k, v in optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = saved_state.get(p, {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
        if self._slots[-1].static_grads is not None:
            for p, g in first.static_grads:
                p.grad = g

    # slot 0 under the names the one-slot step has always had
    batch = property(lambda self: self._slots[0].batch)
    ratings = property(lambda self: self._slots[0].ratings)
    loss = property(lambda self: self._slots[0].loss)
    gnorm = property(lambda self: self._slots[0].gnorm)
    pred = property(lambda self: self._slots[0].pred)
    g_fwd_bwd = property(lambda self: self._slots[0].g_fwd_bwd)
    g_update = property(lambda self: self._slots[0].g_update)
    _flat = property(lambda self: self._slots[0].flat)

    @property
    def slots(self) -> int:
        return len(self._slots)

    def kernel_launches(self, slot: int = 0):
        """Kernel nodes of the recorded step (slot `slot`: forward + backward [+ the update graph of a data-parallel step]): what
        one replay launches.  Needs keep_graph=True at construction; None when the runtime does not hand the graph out."""
        sl = self._slots[slot]
        try:
            return sum(_count_kernel_nodes(g) for g in (sl.g_fwd_bwd, sl.g_update) if g is not None)
        except Exception:
            return None

    def slot_inputs(self, slot: int = 0):
        """(batch views, ratings view, the whole block as uint8) of input slot `slot`: what a loader writes into."""
        sl = self._slots[slot]
        return sl.batch, sl.ratings, sl.flat

    def pack(self, batch, ratings: torch.Tensor) -> torch.Tensor:
        """`batch` + `ratings` as one block in the layout of the step's input buffers (what a loader would stage on the
        device); hand it to __call__(packed=...)."""
        blob = torch.empty_like(self._slots[0].flat)
        for v, src in zip(_flat_views(blob, self._layout, list(batch) + [ratings]), list(batch) + [ratings]):
            v.copy_(src)
        return blob

    def stage(self, slot: int, batch=None, ratings: torch.Tensor | None = None, packed: torch.Tensor | None = None) -> None:
        """Copies a batch into input slot `slot` on the current stream (no replay)."""
        sl = self._slots[slot]
        if packed is not None:
            if packed.shape != sl.flat.shape or packed.dtype != torch.uint8:
                raise RuntimeError("packed batch does not match the step's input layout (use GraphedTrainStep.pack)")
            sl.flat.copy_(packed, non_blocking=True)      # one copy launch for all inputs
        if batch is not None:
            for dst, src in zip(sl.batch, batch):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)
        if ratings is not None and ratings is not sl.ratings:
            sl.ratings.copy_(ratings, non_blocking=True)

    def __call__(self, batch=None, ratings: torch.Tensor | None = None, packed: torch.Tensor | None = None, slot: int = 0):
        """Runs one step on `batch` / `packed` (None: the batch already in the slot's input block).  Returns the slot's
        static (loss, gnorm, pred) tensors -- overwritten by the next replay of that slot."""
        sl = self._slots[slot]
        if batch is not None or ratings is not None or packed is not None:
            self.stage(slot, batch, ratings, packed)
        sl.g_fwd_bwd.replay()
        if sl.static_grads is not None:
            # the graphs read and write the gradient tensors they were recorded with; an eager step in between (a ragged
            # last batch) re-points p.grad elsewhere, so hand the recorded tensors back before anything looks at p.grad
            for p, g in sl.static_grads:
                p.grad = g
        if sl.g_update is not None:
            tap = getattr(self.grad_sync, "tap", None)
            if tap is not None:
                tap.mark_replayed()          # the replayed backward refilled the tap buffers (distributed.TapExchange)
            self.grad_sync(self.model)
            sl.g_update.replay()
        return sl.loss, sl.gnorm, sl.pred
