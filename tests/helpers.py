"""Shared helpers of the GPU parity tests."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import torch


def golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sample(flat, n=4096):
    """Same strided sample as tests/golden/make_golden.py::_sample."""
    flat = np.asarray(flat).reshape(-1)
    if flat.size <= n:
        return flat.copy()
    step = flat.size // n
    return flat[::step][:n].copy()


def max_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) if a.size else 0.0


def check_grads(named_grads, g, rtol=2e-4, atol=2e-6):
    """Each gradient against the golden one: |diff| <= atol + rtol * ||g_ref||_2 elementwise, plus the norm."""
    for k, grad in named_grads.items():
        gn = grad.detach().cpu().numpy()
        ref_l2 = float(g[f"gradl2/{k}"])
        l2 = float(np.sqrt((gn.astype(np.float64) ** 2).sum()))
        assert abs(l2 - ref_l2) <= rtol * ref_l2 + atol, f"grad norm of {k}: {l2} vs {ref_l2}"
        if f"grad/{k}" in g:
            e = max_err(gn, g[f"grad/{k}"])
        else:
            e = max_err(sample(gn), g[f"gradsample/{k}"])
        assert e <= atol + rtol * ref_l2, f"grad {k}: max err {e:.3e}, ref l2 {ref_l2:.3e}"


def check_params_after(model, g, tag, tol_max=1e-3, tol_rms=1e-4):
    """Parameters after 1 / 3 Adam steps.

    Adam normalises every element's step to ~lr whatever the gradient's size, so an element whose
    gradient is itself at rounding-noise level can legitimately move by a fraction of lr = 2e-3
    differently (seen: 3e-4 on 1 of 225k conv weights at the cfg2 shape).  The gate is therefore a
    loose max (lr/2) plus a tight RMS; a wrong gradient is caught by check_grads, which is tight."""
    for k, p in model.named_parameters():
        if float(g[f"gradl2/{k}"]) < 1e-6:
            continue
        v = p.detach().cpu().numpy()
        if f"{tag}/{k}" in g:
            a, b = v, g[f"{tag}/{k}"]
        elif f"{tag}sample/{k}" in g:
            a, b = sample(v), g[f"{tag}sample/{k}"]
        else:
            continue
        diff = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
        assert diff.shape == np.asarray(b).shape
        assert float(np.abs(diff).max()) <= tol_max, f"{tag} {k}: max err {np.abs(diff).max():.3e}"
        assert float(np.sqrt((diff ** 2).mean())) <= tol_rms, f"{tag} {k}: rms err {np.sqrt((diff ** 2).mean()):.3e}"
