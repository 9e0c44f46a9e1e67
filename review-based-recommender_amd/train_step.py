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


def make_optimizer(model: nn.Module, lr: float = LR, fused: bool | None = None) -> torch.optim.Optimizer:
    """torch.optim.Adam(model.parameters(), lr=args.lr)  (train_deepconn_pp.py:135).

    `fused=None` picks torch's single-kernel ("fused") Adam implementation when every parameter lives on
    a HIP device and the default multi-kernel one otherwise; both compute the same update."""
    params = list(model.parameters())
    if fused is None:
        fused = len(params) > 0 and all(p.is_cuda for p in params)
    return torch.optim.Adam(params, lr=lr, fused=True) if fused else torch.optim.Adam(params, lr=lr)


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
