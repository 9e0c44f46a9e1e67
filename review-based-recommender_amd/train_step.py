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

import torch
import torch.nn as nn
import torch.nn.functional as F

LR = 2e-3            # default_deepconn_pp.json:24
MAX_GRAD_NORM = 5.0  # default_deepconn_pp.json:27


def make_optimizer(model: nn.Module, lr: float = LR, fused: bool | None = None,
                   capturable: bool = False) -> torch.optim.Optimizer:
    """torch.optim.Adam(model.parameters(), lr=args.lr)  (train_deepconn_pp.py:135).

    `fused=None` picks torch's single-kernel ("fused") Adam implementation when every parameter lives on
    a HIP device and the default multi-kernel one otherwise; both compute the same update.
    `capturable=True` keeps the step counter on the device so the update can be recorded into a hipGraph
    (GraphedTrainStep)."""
    params = list(model.parameters())
    if fused is None:
        fused = len(params) > 0 and all(p.is_cuda for p in params)
    if fused:
        return torch.optim.Adam(params, lr=lr, fused=True, capturable=capturable)
    return torch.optim.Adam(params, lr=lr, capturable=capturable)


def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, batch, ratings: torch.Tensor,
               max_grad_norm: float = MAX_GRAD_NORM, grad_sync=None):
    """One step.  `batch` is the tuple passed to model(*batch); models returning a tuple
    (NARRE: pred, u_att, i_att) contribute their first element.  `grad_sync(model)` is the
    data-parallel gradient all-reduce hook (None on one GPU).  Returns (loss, gnorm, pred) tensors."""
    optimizer.zero_grad()
    out = model(*batch)
    pred = out[0] if isinstance(out, tuple) else out
    loss = F.mse_loss(pred, ratings)
    loss.backward()
    if grad_sync is not None:
        grad_sync(model)
    gnorm = nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    return loss.detach(), gnorm, pred.detach()


class GraphedTrainStep:
    """train_step() recorded once into a hipGraph and replayed: the step is ~70 short kernels (0.8 ms of GPU
    work at the cfg2 shape), so launching them one by one from Python leaves the GPU waiting on the host.

    Everything data-dependent in the HIP path (distinct-token list, work lists, argmax windows) lives in device
    memory and every launch has a shape-only grid, so a recorded step is valid for any batch of the same shape:
    __call__ copies the new batch into the static input buffers and replays.  With a data-parallel
    `grad_sync` the step is recorded as two graphs (zero_grad+forward+backward | clip+Adam) sharing one memory
    pool, and the RCCL all-reduce runs eagerly between them.

    The optimizer must have been built with make_optimizer(..., capturable=True)."""

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, batch, ratings: torch.Tensor,
                 max_grad_norm: float = MAX_GRAD_NORM, grad_sync=None, warmup: int = 3):
        if not ratings.is_cuda:
            raise RuntimeError("GraphedTrainStep needs HIP tensors")
        self.model, self.optimizer, self.grad_sync, self.max_grad_norm = model, optimizer, grad_sync, max_grad_norm
        self.batch = tuple(t.clone() for t in batch)
        self.ratings = ratings.clone()
        # the warm-up steps below must not count as training: parameters and Adam state are put back in place
        # (same storage -- the recorded graph keeps their addresses) once the graph exists
        saved_params = [p.detach().clone() for p in model.parameters()]
        saved_state = {p: {k: v.clone() for k, v in optimizer.state[p].items() if torch.is_tensor(v)}
                       for group in optimizer.param_groups for p in group["params"] if p in optimizer.state}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up off the default stream: allocator, lazy optimizer state, occupancy queries
            for _ in range(warmup):
                train_step(model, optimizer, self.batch, self.ratings, max_grad_norm, grad_sync)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)
        self.g_fwd_bwd = torch.cuda.CUDAGraph()
        self.g_update = None
        if grad_sync is None:
            with torch.cuda.graph(self.g_fwd_bwd):
                self.loss, self.gnorm, self.pred = train_step(model, optimizer, self.batch, self.ratings, max_grad_norm)
        else:
            with torch.cuda.graph(self.g_fwd_bwd):
                optimizer.zero_grad()
                out = model(*self.batch)
                pred = out[0] if isinstance(out, tuple) else out
                loss = F.mse_loss(pred, self.ratings)
                loss.backward()
                self.loss, self.pred = loss.detach(), pred.detach()
            grad_sync(model)
            self.g_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_update, pool=self.g_fwd_bwd.pool()):
                self.gnorm = nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
                optimizer.step()
        with torch.no_grad():
            for p, v in zip(model.parameters(), saved_params):
                p.copy_(v)
            for group in optimizer.param_groups:
                for p in group["params"]:
                    for k, v in optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = saved_state.get(p, {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()

    def __call__(self, batch=None, ratings: torch.Tensor | None = None):
        """Runs one step on `batch` (None: the batch already in the static buffers).  Returns the graph's
        static (loss, gnorm, pred) tensors -- overwritten by the next replay."""
        if batch is not None:
            for dst, src in zip(self.batch, batch):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)
        if ratings is not None and ratings is not self.ratings:
            self.ratings.copy_(ratings, non_blocking=True)
        self.g_fwd_bwd.replay()
        if self.g_update is not None:
            self.grad_sync(self.model)
            self.g_update.replay()
        return self.loss, self.gnorm, self.pred
