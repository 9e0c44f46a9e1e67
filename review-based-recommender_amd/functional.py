"""torch.autograd.Functions over the C ABI (include/rbr_hip.h).  HIP tensors only."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ACT_RELU, ACT_TANH, PAD_SAME, PAD_VALID, TIMER, check, current_stream, dev_ptr, ptr_array

F32, I64, I32, U8 = torch.float32, torch.int64, torch.int32, torch.uint8


def _mask_u8(mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    if mask.dtype == torch.uint8:
        return mask.contiguous()
    raise RuntimeError(f"mask must be bool or uint8, got {mask.dtype}")


# Data-parallel tap exchange (distributed.TapExchange): when a sink is installed, the backward of an un-gated conv over the
# sink's word table emits its (token, value) taps instead of the dense table gradient; the sink rebuilds the averaged
# gradient of all ranks after an all-gather of the taps (rbr_textcnn_bwd_taps / rbr_textcnn_dtable_from_taps).
_TAP_SINKS: dict = {}       # word-table data_ptr -> sink: several models (an eval / EMA copy, a second trainer) can coexist


def set_prod_precision(name: Optional[str]) -> None:
    """Arithmetic of the token-product conv's GEMM (rbr_set_prod_precision): "f32" (f32 MFMA, an fma chain bit for bit),
    "bf16x3" (default: exact three-plane bf16 split, 6 plane products, f32 accumulation: f32-class accuracy), "bf16x2",
    "bf16" (operands rounded to bf16: the reduced-precision row of BASELINE configs 3 and 5); None restores the default.
    Set it between steps, never between a forward and its backward (the workspace layout depends on it)."""
    if name is not None and name not in _lib.PROD_PRECISIONS:
        raise ValueError(f"unknown product precision {name!r}; one of {sorted(_lib.PROD_PRECISIONS)}")
    _lib.lib().rbr_set_prod_precision(-1 if name is None else _lib.PROD_PRECISIONS[name])


_DTABLE_MODE: list = [None]


def set_dtable_mode(mode: Optional[str]) -> None:
    """How an un-gated conv's backward accumulates the word-table gradient: None (default; env RBR_DTABLE_MODE) = G built with
    f32 atomics, whose arrival order leaves rounding-level run-to-run noise in the result; "fixed" = every (token, tap, channel)
    cell summed in 64-bit fixed point (2^-40 units, integer adds: order-free) -- two runs over the same batch give the same
    bits, like the reference's CPU embedding backward (models/deepconn/layers.py:22-24).  It is the data-parallel tap rebuild run
    on this rank's taps alone (rbr_textcnn_bwd_taps + rbr_textcnn_dtable_from_taps, n_sets = 1): a sort per step, eager
    launches only (rocPRIM's sort cannot be recorded into a hipGraph here), D % 4 == 0."""
    if mode not in (None, "fixed"):
        raise ValueError(f"unknown dtable mode {mode!r}; None or 'fixed'")
    _DTABLE_MODE[0] = mode


def _dtable_fixed() -> bool:
    return (_DTABLE_MODE[0] or os.environ.get("RBR_DTABLE_MODE")) == "fixed"


def get_prod_precision() -> str:
    mode = _lib.lib().rbr_get_prod_precision()
    return {v: k for k, v in _lib.PROD_PRECISIONS.items()}[mode]


# --------------------------------------------------------------------------- id range check
_ID_ERR: dict = {}       # per device: int64[4] error record of rbr_sanitize_ids (zero while clean)


def _id_err(dev) -> torch.Tensor:
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _ID_ERR.get(idx)
    if t is None:
        t = _ID_ERR[idx] = torch.zeros(4, dtype=torch.int64, device=dev)
    return t


def sanitize_ids(sets, stack_first_two: bool = False):
    """Range check in front of the embedding-style lookups: `sets` = [(ids int64 tensor, table rows, replacement row), ...]
    (at most 8).  Returns clean copies (one launch, one allocation): an id outside [0, rows) becomes `replacement` and is
    recorded on the device; check_id_errors() raises the IndexError nn.Embedding would have raised (reference
    models/deepconn/layers.py:23) at the caller's next synchronisation point.  With `stack_first_two` the first two
    outputs (equal shapes) are adjacent and the first return value is the stacked [2n, ...] tensor (user rows first)."""
    if not sets or len(sets) > _lib.MAX_ID_SETS:
        raise RuntimeError(f"sanitize_ids takes 1..{_lib.MAX_ID_SETS} id tensors")
    dev = sets[0][0].device
    tens = []
    for t, rows, rep in sets:
        if t.dtype != I64:
            raise RuntimeError(f"ids must be int64, got {t.dtype}")
        dev_ptr(t.contiguous(), I64, "ids")        # device gate
        tens.append(t.contiguous())
    flat = torch.empty(sum(t.numel() for t in tens), dtype=I64, device=dev)
    outs, o = [], 0
    arr = (_lib.IdSet * len(tens))()
    for k, (t, (_, rows, rep)) in enumerate(zip(tens, sets)):
        v = flat[o:o + t.numel()].view(t.shape)
        outs.append(v)
        o += t.numel()
        rep = 0 if rep is None or rep < 0 else int(rep)
        arr[k] = _lib.IdSet(t.data_ptr() if t.numel() else None, v.data_ptr() if t.numel() else None, t.numel(), int(rows), rep)
    check(_lib.lib().rbr_sanitize_ids(len(tens), arr, _id_err(dev).data_ptr(), current_stream()), "rbr_sanitize_ids")
    if stack_first_two:
        a, b = tens[0], tens[1]
        if a.shape != b.shape:
            raise RuntimeError("stack_first_two needs equal shapes")
        outs = [flat[:2 * a.numel()].view(2 * a.shape[0], *a.shape[1:])] + outs[2:]
    return outs


def check_id_errors(device=None) -> None:
    """Raises IndexError when any id seen by sanitize_ids() since the last call was outside its table (and clears the
    record).  Synchronises with the device: call it where the trainer synchronises anyway (its loss.item() log points)."""
    devs = list(_ID_ERR) if device is None else [torch.device(device).index if torch.device(device).index is not None
                                                 else torch.cuda.current_device()]
    for idx in devs:
        t = _ID_ERR.get(idx)
        if t is None:
            continue
        rec = t.cpu()
        if int(rec[0]) > 0:
            t.zero_()
            raise IndexError(f"index out of range in self: {int(rec[0])} id(s) outside their table on cuda:{idx}, e.g. "
                             f"{int(rec[1])} in id tensor #{int(rec[2])} of its call; such ids were replaced by the padding row")


def dedup_rows(u_ids: torch.Tensor, i_ids: torch.Tensor, user_size: int, item_size: int, masks: Optional[torch.Tensor], L: int):
    """In-batch dedup of the documents by id (rbr_dedup_rows): returns (first [2B] int64, masks_out [2B, L] bool).  Rows
    [0, B) are the user side, [B, 2B) the item side; first[r] is the first row of the same side carrying the same id, and
    masks_out blanks every row that is not its own first occurrence, so the encoder skips the repeated documents.
    Everything has a static shape and nothing synchronises: the step stays graph-capturable."""
    B = u_ids.shape[0]
    dev = u_ids.device
    u_ids, i_ids = u_ids.contiguous(), i_ids.contiguous()
    mask8 = _mask_u8(masks)
    if mask8 is not None and mask8.shape != (2 * B, L):
        raise RuntimeError(f"masks must be [{2 * B}, {L}], got {tuple(mask8.shape)}")
    L_ = _lib.lib()
    ws = torch.empty(L_.rbr_dedup_ws_bytes(int(user_size), int(item_size)), dtype=torch.uint8, device=dev)
    first = torch.empty(2 * B, dtype=I64, device=dev)
    out = torch.empty(2 * B, L, dtype=torch.uint8, device=dev)
    check(L_.rbr_dedup_rows(B, int(L), dev_ptr(u_ids, I64, "u_ids"), dev_ptr(i_ids, I64, "i_ids"), int(user_size), int(item_size),
                            dev_ptr(mask8, U8, "mask"), ws.data_ptr(), dev_ptr(first, I64, "first"), dev_ptr(out, U8, "mask_out"),
                            current_stream()), "rbr_dedup_rows")
    return first, out.view(torch.bool)


def set_tap_sink(sink, table: Optional[torch.Tensor] = None) -> None:
    """Installs `sink` for the word table it exchanges (sink.table); set_tap_sink(None, table) removes that table's sink,
    set_tap_sink(None) removes all of them."""
    if sink is None:
        if table is None:
            _TAP_SINKS.clear()
        else:
            _TAP_SINKS.pop(table.data_ptr(), None)
        return
    _TAP_SINKS[sink.table.data_ptr()] = sink


# ---- one gradient buffer for several producers of a table's gradient ------------------------------------------------------
def _fanout_acc(table: torch.Tensor):
    """Called in an op's forward: the (buffer, stream) of the table_fanout `table` is an alias of, or None.  The buffer hangs on
    the fan-out's own autograd node, so concurrent or interleaved forwards / backwards of several steps cannot meet in one."""
    node = table.grad_fn
    return getattr(node, "acc", None) if type(node).__name__ == "_TableFanoutBackward" else None


def _table_acc(acc, device):
    """In an op's backward: the shared buffer when this op may add its rows to it (same stream as the fan-out), else None."""
    if acc is None or acc[1] != torch.cuda.current_stream(device):
        return None
    return acc[0]


class _TableFanout(torch.autograd.Function):
    """n aliases of a table, one per consumer.  The consumers' backwards that support it add their rows into ONE zeroed buffer
    (kept on this node) and each hand that same buffer back; here the duplicates are dropped, so a step with 8 producers of
    table gradient (D-ATT: two gates and two convs per tower) pays one 20 MB fill instead of seven 60 MB autograd adds and eight
    dense [V, D] outputs.  A consumer that returns a tensor of its own is added the ordinary way."""

    @staticmethod
    def forward(ctx, table, n):
        ctx.set_materialize_grads(False)
        ctx.acc = (torch.zeros_like(table), torch.cuda.current_stream(table.device))
        return tuple(table.view(table.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        total, seen = None, set()
        for g in grads:
            if g is None or g.data_ptr() in seen:
                continue
            seen.add(g.data_ptr())
            total = g if total is None else total + g
        return total, None


def table_fanout(table: torch.Tensor, n: int):
    """`n` aliases of `table` for `n` consumer ops of one step (hand each op its own alias).  With gradients on, the ops that
    can (token-product convs and gates) then add their table-gradient rows into one shared buffer instead of producing a
    dense gradient each: see _TableFanout.  Without gradients: the table itself, n times."""
    if not (torch.is_grad_enabled() and table.requires_grad and table.is_cuda):
        return (table,) * n
    return _TableFanout.apply(table, n)


class _TextCNN(torch.autograd.Function):
    """feat[n_docs, C] = pool(act(conv(mask * gate * table[ids])))  -- see rbr_textcnn_* in rbr_hip.h."""

    @staticmethod
    def forward(ctx, table, gate, ids, mask, kernel_sizes, pad_mode, act, padding_idx, conv_flags, *wb):
        ctx.fanout_acc = _fanout_acc(table)
        n = len(kernel_sizes)
        weights, biases = wb[:n], wb[n:]
        dev_ptr(table.contiguous(), F32, "word table")   # device / dtype gate before anything touches HIP
        if ids.dim() != 2:
            raise RuntimeError("ids must be [n_docs, L]")
        n_docs, L = ids.shape
        V, D = table.shape
        for k, w, b in zip(kernel_sizes, weights, biases):
            assert w.shape[1] == D and w.shape[2] == k and b.shape[0] == w.shape[0]
        channels = [w.shape[0] for w in weights]
        Ctot = sum(channels)
        desc = _lib.make_desc(n_docs, L, D, V, kernel_sizes, channels, pad_mode, act, padding_idx, conv_flags)
        L_ = _lib.lib()
        dev = table.device
        ids = ids.contiguous()
        mask8 = _mask_u8(mask)
        if mask8 is not None and mask8.shape != ids.shape:
            raise AssertionError("inputs.shape[:-1] == masks.shape")   # deepconn/utils.py:58
        if gate is not None:
            gate = gate.contiguous()
        table_c = table.contiguous()
        ws = [w.contiguous() for w in weights]
        bs = [b.contiguous() for b in biases]

        n_packed = L_.rbr_textcnn_packed_floats(C.byref(desc))
        n_part = L_.rbr_textcnn_partial_elems(C.byref(desc))
        if n_packed == 0 or n_part == 0:
            check(-1, "rbr_textcnn plan")
        pval = torch.empty(n_part, dtype=F32, device=dev)
        pidx = torch.empty(n_part, dtype=I32, device=dev)
        feat = torch.empty(n_docs, Ctot, dtype=F32, device=dev)
        argmax = torch.empty(n_docs, Ctot, dtype=I32, device=dev)
        st = current_stream()

        ws_bytes = L_.rbr_textcnn_fwd_ws_bytes(C.byref(desc))      # > 0: the token-product formulation will run
        prod_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        packed = None         # MFMA tile image of the conv weights: the dense forward and the scatter backward read it
        if prod_ws is None:
            packed = _TextCNN._pack(L_, desc, ws, n_packed, dev, st)
        if prod_ws is not None:
            # token-product formulation, stage by stage (each stage is what rbr_textcnn_conv_fwd would run)
            wsp = prod_ws.data_ptr()
            ev = TIMER.record("textcnn_prod_prepare")
            check(L_.rbr_textcnn_prod_prepare(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                              ptr_array(ws, F32, "conv weight"), dev_ptr(pidx, I32, "pidx"), wsp, st),
                  "rbr_textcnn_prod_prepare")
            if ev is not None:
                ev.record()
            ev = TIMER.record("textcnn_prod_table")
            check(L_.rbr_textcnn_prod_table(C.byref(desc), dev_ptr(table_c, F32, "word table"), wsp, st),
                  "rbr_textcnn_prod_table")
            if ev is not None:
                ev.record()
            ev = TIMER.record("textcnn_prod_pool")
            check(L_.rbr_textcnn_prod_pool(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                           dev_ptr(gate, F32, "gate"), dev_ptr(pval, F32, "pval"), dev_ptr(pidx, I32, "pidx"),
                                           wsp, st), "rbr_textcnn_prod_pool")
            if ev is not None:
                ev.record()
        else:
            ev = TIMER.record("textcnn_conv_fwd")
            check(L_.rbr_textcnn_conv_fwd(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                          dev_ptr(gate, F32, "gate"), dev_ptr(table_c, F32, "word table"),
                                          ptr_array(ws, F32, "conv weight"), dev_ptr(packed, F32, "packed"),
                                          dev_ptr(pval, F32, "pval"), dev_ptr(pidx, I32, "pidx"), None, st),
                  "rbr_textcnn_conv_fwd")
            if ev is not None:
                ev.record()
        check(L_.rbr_textcnn_pool_finalize(C.byref(desc), dev_ptr(pval, F32, "pval"), dev_ptr(pidx, I32, "pidx"),
                                           ptr_array(bs, F32, "conv bias"), dev_ptr(feat, F32, "feat"),
                                           dev_ptr(argmax, I32, "argmax"), st), "rbr_textcnn_pool_finalize")
        ctx.desc = desc
        ctx.n = n
        ctx.has_gate = gate is not None
        ctx.has_mask = mask8 is not None
        ctx.prod_ws = prod_ws          # distinct-token list of the batch, reused by the table-gradient GEMM
        ctx.save_for_backward(table_c, ids, packed, feat, argmax, *([mask8] if mask8 is not None else []),
                              *([gate] if gate is not None else []), *ws)
        ctx.mark_non_differentiable(argmax)
        ctx.set_materialize_grads(False)        # no zero-filled "gradient" tensor for argmax (a fill launch per step)
        return feat, argmax

    @staticmethod
    def _pack(L_, desc, ws, n_packed, dev, st):
        packed = torch.empty(n_packed, dtype=F32, device=dev)
        check(L_.rbr_textcnn_pack(C.byref(desc), ptr_array(ws, F32, "conv weight"), dev_ptr(packed, F32, "packed"), st),
              "rbr_textcnn_pack")
        return packed

    @staticmethod
    def backward(ctx, d_feat, _d_argmax):
        saved = list(ctx.saved_tensors)
        table, ids, packed, feat, argmax = saved[:5]
        k = 5
        mask8 = None
        gate = None
        if ctx.has_mask:
            mask8 = saved[k]; k += 1
        if ctx.has_gate:
            gate = saved[k]; k += 1
        S = _ConvSaved(table=table, ids=ids, packed=packed, feat=feat, argmax=argmax, mask8=mask8, gate=gate, ws=list(saved[k:]),
                       desc=ctx.desc, prod_ws=ctx.prod_ws, fanout_acc=ctx.fanout_acc, bws=None)
        dtable, dgate, dWs, dbs = _textcnn_backward(S, d_feat, ctx.needs_input_grad[0], ctx.has_gate and ctx.needs_input_grad[1])
        return (dtable, dgate, None, None, None, None, None, None, None, *dWs, *dbs)


class _ConvSaved:
    """What a conv forward leaves for its backward (a plain record: the fused encoder + head function shares the backward)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _textcnn_backward(S, d_feat, need_table: bool, need_gate: bool):
    """Backward of the fused encoder from d_feat [n_docs, C]: returns (dtable, dgate, dWs, dbs).  dtable is None when the table
    gradient left through a side channel instead: the data-parallel tap sink, or -- compact row form -- the optimizer that
    registered for it (set_row_grad_sink).  S.bws: the G workspace the forward allocated and cleared (or None)."""
    table, ids, packed, feat, argmax, mask8, gate, ws, desc = (S.table, S.ids, S.packed, S.feat, S.argmax, S.mask8, S.gate,
                                                               S.ws, S.desc)
    if d_feat is None:
        d_feat = torch.zeros_like(feat)
    L_ = _lib.lib()
    dev = table.device
    d_feat = d_feat.contiguous()
    dWs = [torch.empty_like(w) for w in ws]
    dbs = [torch.empty(w.shape[0], dtype=F32, device=dev) for w in ws]
    sink = _TAP_SINKS.get(table.data_ptr())          # only the sink installed for THIS table ever sees the call
    use_taps = (sink is not None and need_table and gate is None and sink.accepts(table, desc, L_))
    if use_taps:
        need_table = False          # table.grad is produced by the exchange, after the all-gather of the taps
    fixed_dtable = None
    if need_table and gate is None and _dtable_fixed():
        fixed_bytes = L_.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(desc), 1)
        if fixed_bytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("RBR_DTABLE_MODE=fixed sorts with rocPRIM and cannot be recorded into a hipGraph: run the step eagerly")
            n_taps = L_.rbr_textcnn_taps_count(C.byref(desc))
            tok = torch.empty(n_taps, dtype=I32, device=dev)
            val = torch.empty(n_taps, dtype=F32, device=dev)
            tws = torch.empty(fixed_bytes, dtype=torch.uint8, device=dev)
            fixed_dtable = torch.empty_like(table)
            check(L_.rbr_textcnn_bwd_taps(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"), dev_ptr(feat, F32, "feat"),
                                          dev_ptr(argmax, I32, "argmax"), dev_ptr(d_feat, F32, "d_feat"), dev_ptr(tok, I32, "tap tokens"),
                                          dev_ptr(val, F32, "tap values"), current_stream()), "rbr_textcnn_bwd_taps")
            check(L_.rbr_textcnn_dtable_from_taps(C.byref(desc), 1, dev_ptr(tok, I32, "tap tokens"), dev_ptr(val, F32, "tap values"),
                                                  ptr_array(ws, F32, "conv weight"), tws.data_ptr(), dev_ptr(fixed_dtable, F32, "dtable"),
                                                  current_stream()), "rbr_textcnn_dtable_from_taps")
            need_table = False
    bws_bytes = L_.rbr_textcnn_bwd_prod_ws_bytes(C.byref(desc)) if S.prod_ws is not None else 0
    # dense forward (no token list of its own), un-gated: the table gradient still goes through a distinct-token list built
    # here (16 us) instead of the window scatter's row of f32 atomics per (document, channel, tap)
    list_bytes = (L_.rbr_textcnn_bwd_dtable_list_ws_bytes(C.byref(desc))
                  if (need_table and not bws_bytes and gate is None) else 0)
    acc = _table_acc(S.fanout_acc, dev) if (need_table and bws_bytes) else None
    # compact row gradient: the optimizer that asked for it takes the rows of the batch's tokens, nobody writes (or later
    # reads) the zero rows of a dense [V, D] gradient
    row_sink = _row_grad_sink_for(table) if (need_table and bws_bytes and gate is None and acc is None) else None
    # the token-list backwards overwrite the whole table gradient; the window scatter accumulates into zeros
    dtable = None
    if need_table and row_sink is None:
        dtable = torch.empty_like(table) if (bws_bytes or list_bytes) else torch.zeros_like(table)
    # the token-product backward zeroes dgate in the launch that zeroes its G rows; the window scatter accumulates into zeros
    dgate = (torch.empty_like(gate) if bws_bytes else torch.zeros_like(gate)) if need_gate else None
    wsn = L_.rbr_textcnn_bwd_ws_floats(C.byref(desc))
    wsb = torch.empty(max(wsn, 1), dtype=F32, device=dev)
    st = current_stream()
    common = (C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"), dev_ptr(gate, F32, "gate"),
              dev_ptr(table, F32, "table"))
    # many short documents (NARRE's reviews): dW = G^T @ table[distinct tokens] on the MFMA pipe, from the G the
    # table-gradient call builds anyway (0 floats of workspace = not this shape: the window-row kernels below)
    dwg_floats = L_.rbr_textcnn_bwd_dw_from_g_ws_floats(C.byref(desc)) if (need_table and bws_bytes) else 0
    join = None
    fork = None
    side = None
    if not dwg_floats:
        # the weight-gradient kernels and the table-gradient kernels below both start from d_feat and share nothing
        # else: the former go to a second stream (fork here, join before returning), so the two chains overlap --
        # also inside a captured graph, where the fork becomes two parallel branches
        side = _side_stream(dev) if (need_table or need_gate or use_taps) else None
        if side is not None:
            fork = torch.cuda.Event()
            fork.record()

    def run_dw():
        """The weight-gradient chain, enqueued AFTER the table-gradient chain: in a replayed graph the branch captured first
        keeps the queue of the nodes before and after the fork, and the join of the other branch into that queue costs
        ~10 us -- so the longer (table) branch goes first and the step's next kernel follows it without a gap."""
        if dwg_floats:
            return join
        j = None
        if side is not None:
            side.wait_event(fork)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            ev_dw = TIMER.record("textcnn_bwd_dw")
            check(L_.rbr_textcnn_bwd_dw(*common, dev_ptr(feat, F32, "feat"), dev_ptr(argmax, I32, "argmax"),
                                        dev_ptr(d_feat, F32, "d_feat"), ptr_array(dWs, F32, "dW"),
                                        ptr_array(dbs, F32, "dbias"), dev_ptr(wsb, F32, "ws"), current_stream()),
                  "rbr_textcnn_bwd_dw")
            if ev_dw is not None:
                ev_dw.record()
            if side is not None:
                j = torch.cuda.Event()
                j.record()
        return j

    ev = TIMER.record("textcnn_bwd_dtable")
    if fixed_dtable is not None:          # table gradient done above (order-free fixed-point sums); the weight gradient remains
        if ev is not None:
            ev.record()
        _join(run_dw())
        return fixed_dtable, dgate, dWs, dbs
    if use_taps:
        tok, val = sink.local_buffers(L_.rbr_textcnn_taps_count(C.byref(desc)), dev)
        check(L_.rbr_textcnn_bwd_taps(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                      dev_ptr(feat, F32, "feat"), dev_ptr(argmax, I32, "argmax"),
                                      dev_ptr(d_feat, F32, "d_feat"), dev_ptr(tok, I32, "tap tokens"),
                                      dev_ptr(val, F32, "tap values"), st), "rbr_textcnn_bwd_taps")
        sink.record(desc, list(ws))
        if ev is not None:
            ev.record()
        _join(run_dw())
        return None, dgate, dWs, dbs
    if (need_table or need_gate) and bws_bytes:
        # token-product backward: dtable = G @ Wprod^T over the forward's distinct-token list (no atomics on the
        # table); d(gate) of gated convs (D-ATT) is read off the forward's product table
        zeroed = S.bws is not None                    # the forward's gather launch cleared G's rows
        bws = S.bws if zeroed else torch.empty(bws_bytes, dtype=torch.uint8, device=dev)
        flags = _lib.G_ZEROED if zeroed else 0
        rows = sq = None
        if row_sink is not None:
            cap = C.c_int32(0)
            rot = C.c_void_p()
            check(L_.rbr_textcnn_token_list(C.byref(desc), S.prod_ws.data_ptr(), C.byref(rot), None, None, C.byref(cap)),
                  "rbr_textcnn_token_list")
            rows = torch.empty(cap.value, table.shape[1], dtype=F32, device=dev)
            sq = torch.empty(L_.rbr_textcnn_row_grad_partials(C.byref(desc)), dtype=F32, device=dev)   # one per workgroup
            flags |= _lib.G_ROWS
        out = rows if rows is not None else dtable
        if dwg_floats:
            # G first; then its two consumers side by side: dtable = G @ Wprod^T on this stream, dW = G^T @ table rows on the
            # second one (fork after the build, join before returning; two parallel branches in a captured graph)
            check(L_.rbr_textcnn_bwd_dtable_prod_ex(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                                    dev_ptr(gate, F32, "gate"), dev_ptr(feat, F32, "feat"),
                                                    dev_ptr(argmax, I32, "argmax"), dev_ptr(d_feat, F32, "d_feat"),
                                                    S.prod_ws.data_ptr(), bws.data_ptr(), None, dev_ptr(dgate, F32, "dgate"), None,
                                                    _lib.G_BUILD | (flags & _lib.G_ZEROED), st), "rbr_textcnn_bwd_g_build")
            dwg_ws = torch.empty(dwg_floats, dtype=F32, device=dev)
            side = _side_stream(dev)
            if side is not None:
                fork = torch.cuda.Event()
                fork.record()
                side.wait_event(fork)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                ev2 = TIMER.record("textcnn_bwd_dw")
                check(L_.rbr_textcnn_bwd_dw_from_g(C.byref(desc), dev_ptr(table, F32, "table"), dev_ptr(feat, F32, "feat"),
                                                   dev_ptr(d_feat, F32, "d_feat"), S.prod_ws.data_ptr(), bws.data_ptr(),
                                                   ptr_array(dWs, F32, "dW"), ptr_array(dbs, F32, "dbias"),
                                                   dev_ptr(dwg_ws, F32, "ws"), current_stream()), "rbr_textcnn_bwd_dw_from_g")
                if ev2 is not None:
                    ev2.record()
                if side is not None:
                    join = torch.cuda.Event()
                    join.record()
            if need_table:
                ev3 = TIMER.record("textcnn_bwd_g_product")
                check(L_.rbr_textcnn_bwd_dtable_prod_ex(C.byref(desc), None, None, None, None, None, None, S.prod_ws.data_ptr(),
                                                        bws.data_ptr(), dev_ptr(out, F32, "dtable"), None, dev_ptr(sq, F32, "sq_part"),
                                                        _lib.G_PRODUCT | (flags & _lib.G_ROWS), st), "rbr_textcnn_bwd_g_product")
                if ev3 is not None:
                    ev3.record()
            if ev is not None:
                ev.record()
            _join(join)
        else:
            if acc is not None:          # shared gradient buffer of the step (table_fanout): rows added, buffer handed back
                dtable = out = acc
                flags |= _lib.G_ACCUMULATE
            if ev is not None and out is not None:
                # timing pass (bench.py): the call's two phases as two calls, so that the sparse product (ONE launch,
                # g_times_w) gets HIP events of its own -- same kernels, same order, same arguments
                check(L_.rbr_textcnn_bwd_dtable_prod_ex(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                                        dev_ptr(gate, F32, "gate"), dev_ptr(feat, F32, "feat"),
                                                        dev_ptr(argmax, I32, "argmax"), dev_ptr(d_feat, F32, "d_feat"),
                                                        S.prod_ws.data_ptr(), bws.data_ptr(), None, dev_ptr(dgate, F32, "dgate"),
                                                        None, _lib.G_BUILD | (flags & _lib.G_ZEROED), st), "rbr_textcnn_bwd_g_build")
                ev.record()
                ev = TIMER.record("textcnn_bwd_g_product")
                check(L_.rbr_textcnn_bwd_dtable_prod_ex(C.byref(desc), None, None, None, None, None, None, S.prod_ws.data_ptr(),
                                                        bws.data_ptr(), dev_ptr(out, F32, "dtable"), None, dev_ptr(sq, F32, "sq_part"),
                                                        _lib.G_PRODUCT | (flags & (_lib.G_ROWS | _lib.G_ACCUMULATE)), st),
                      "rbr_textcnn_bwd_g_product")
                ev.record()
                ev = None
            else:
                check(L_.rbr_textcnn_bwd_dtable_prod_ex(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                                        dev_ptr(gate, F32, "gate"), dev_ptr(feat, F32, "feat"),
                                                        dev_ptr(argmax, I32, "argmax"), dev_ptr(d_feat, F32, "d_feat"),
                                                        S.prod_ws.data_ptr(), bws.data_ptr(), dev_ptr(out, F32, "dtable"),
                                                        dev_ptr(dgate, F32, "dgate"), dev_ptr(sq, F32, "sq_part"),
                                                        _lib.G_BUILD | _lib.G_PRODUCT | flags, st), "rbr_textcnn_bwd_dtable_prod")
            if ev is not None:
                ev.record()
            _join(run_dw())
        if rows is not None:
            row_sink.put_row_grad(table, RowGradient(table, rows, sq, rot.value, S.prod_ws))
            return None, dgate, dWs, dbs
        return dtable, dgate, dWs, dbs
    if list_bytes:
        lws = torch.empty(list_bytes, dtype=torch.uint8, device=dev)
        check(L_.rbr_textcnn_bwd_dtable_list(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                             ptr_array(ws, F32, "conv weight"), dev_ptr(feat, F32, "feat"),
                                             dev_ptr(argmax, I32, "argmax"), dev_ptr(d_feat, F32, "d_feat"), lws.data_ptr(),
                                             dev_ptr(dtable, F32, "dtable"), st), "rbr_textcnn_bwd_dtable_list")
        if ev is not None:
            ev.record()
        _join(run_dw())
        return dtable, dgate, dWs, dbs
    if packed is None:
        packed = _TextCNN._pack(L_, desc, ws, L_.rbr_textcnn_packed_floats(C.byref(desc)), dev, st)
    check(L_.rbr_textcnn_bwd_dtable(*common, dev_ptr(packed, F32, "packed"), dev_ptr(feat, F32, "feat"),
                                    dev_ptr(argmax, I32, "argmax"), dev_ptr(d_feat, F32, "d_feat"),
                                    dev_ptr(dtable, F32, "dtable"), dev_ptr(dgate, F32, "dgate"), st),
          "rbr_textcnn_bwd_dtable")
    if ev is not None:
        ev.record()
    _join(run_dw())
    return dtable, dgate, dWs, dbs


# ---- compact row gradient of an embedding table (consumer: train_step.HipClipAdam) ------------------------------------------
# word-table data_ptr -> weakref of the sink: wants_row_grad(table) -> bool, put_row_grad(table, RowGradient).  Weak: a registry
# entry never keeps an optimizer (and with it the parameters and Adam state it holds) alive, and a sink that was dropped
# without being unregistered simply disappears (the backward then produces the dense gradient again).
_ROW_GRAD_SINKS: dict = {}


def set_row_grad_sink(table: torch.Tensor, sink) -> None:
    """Registers `sink` (None: removes it) for the gradient of `table` in compact row form.  A conv backward over `table` hands
    sink.put_row_grad(table, RowGradient) the rows of the batch's tokens and returns no dense [V, D] gradient ONLY while
    sink.wants_row_grad(table) answers True for that backward -- the sink says per step whether it is the one that will consume
    the gradient (train_step.HipClipAdam: inside its row_grad_scope(), which train_step() / GraphedTrainStep open around the
    forward + backward of a step this optimizer drives).  Every other backward gets the ordinary dense gradient."""
    if sink is None:
        _ROW_GRAD_SINKS.pop(table.data_ptr(), None)
    else:
        _ROW_GRAD_SINKS[table.data_ptr()] = weakref.ref(sink)


def _row_grad_sink_for(table: torch.Tensor):
    """The live sink that wants THIS backward's gradient of `table` in row form, or None."""
    ref = _ROW_GRAD_SINKS.get(table.data_ptr())
    if ref is None:
        return None
    sink = ref()
    if sink is None:                                   # the optimizer is gone: forget the entry
        _ROW_GRAD_SINKS.pop(table.data_ptr(), None)
        return None
    return sink if sink.wants_row_grad(table) else None


class RowGradient:
    """Gradient of an embedding table as the rows of the tokens one batch holds (rbr_textcnn_bwd_dtable_prod_ex, RBR_G_ROWS):
    rows [cap, D] (row r = token tok_of_row[r] of the forward's list), sq [partials] = sums of squares of the rows, and the
    list's inverse map row_of_token [V] inside the forward's workspace (kept alive here).  Every other row of the dense
    gradient nn.Embedding's backward would build is exactly zero."""

    def __init__(self, table, rows, sq, row_of_token_ptr, keep_alive):
        self.V, self.D = int(table.shape[0]), int(table.shape[1])
        self.rows, self.sq, self.row_of_token_ptr, self._keep = rows, sq, int(row_of_token_ptr), keep_alive

    def to_dense(self) -> torch.Tensor:
        dense = torch.empty(self.V, self.D, dtype=F32, device=self.rows.device)
        check(_lib.lib().rbr_row_grad_to_dense(self.V, self.D, self.row_of_token_ptr, dev_ptr(self.rows, F32, "rows"),
                                               dev_ptr(dense, F32, "dense"), current_stream()), "rbr_row_grad_to_dense")
        return dense


_SIDE_STREAMS: dict = {}


def _side_stream(dev):
    """The second stream of `dev` for the weight-gradient chain (None: RBR_BWD_OVERLAP=0)."""
    if os.environ.get("RBR_BWD_OVERLAP", "1") == "0":
        return None
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
        _lib.FORKED_STREAMS.add(s.cuda_stream)          # forked and joined with events by its users: fine inside a capture
    return s


def _join(event) -> None:
    if event is not None:
        torch.cuda.current_stream().wait_event(event)


def textcnn(table: torch.Tensor, ids: torch.Tensor, mask: Optional[torch.Tensor], weights: Sequence[torch.Tensor],
            biases: Sequence[torch.Tensor], *, gate=None, pad_mode: int = PAD_SAME,
            act: int = ACT_RELU, padding_idx: Optional[int] = 0, return_argmax: bool = False, pad_runs: bool = False,
            gate_split: int = 0):
    """Fused WordEmbedding -> masked_tensor -> MyConv1d -> act -> MaxPool1d(seq_len).

    table [V,D] f32; ids [n_docs,L] int64; mask [n_docs,L] bool or None; weights[w] [C_w,D,kz_w];
    biases[w] [C_w].  Returns feat [n_docs, sum C_w] (width-major channels).
    pad_runs (un-masked convs only): the caller vouches that `gate` is the same at every position whose tokens are all
    padding_idx within 8 positions either side (RBR_CONV_PAD_RUNS): runs of padding are then encoded once, exactly."""
    kernel_sizes = tuple(int(w.shape[2]) for w in weights)
    flags = _lib.CONV_PAD_RUNS if (pad_runs and mask is None and padding_idx is not None
                                   and os.environ.get("RBR_PAD_RUNS", "1") != "0") else 0
    if gate_split:
        # two gates (RBR_CONV_GATE_SPLIT): banks [0, gate_split) under gate[0], the rest under gate[1]; token-product path only
        if not (isinstance(gate, (tuple, list)) and len(gate) == 2 and 0 < gate_split < len(weights)):
            raise RuntimeError("gate_split needs gate=(gate_a, gate_b) and 0 < gate_split < number of banks")
        gate = torch.stack([gate[0], gate[1]])          # [2, n_docs, L]; its backward hands each gate its plane
        flags |= _lib.conv_gate_split(gate_split)
    feat, argmax = _TextCNN.apply(table, gate, ids, mask, kernel_sizes, pad_mode, act, padding_idx, flags, *weights, *biases)
    return (feat, argmax) if return_argmax else feat


def textcnn_product_applies(table: torch.Tensor, ids: torch.Tensor, weights: Sequence[torch.Tensor], pad_mode: int, act: int,
                            padding_idx: Optional[int]) -> bool:
    """True when textcnn() over these shapes runs the token-product formulation, forward AND table-gradient backward (the only
    one that takes a split gate)."""
    V, D = table.shape
    desc = _lib.make_desc(ids.shape[0], ids.shape[1], D, V, [int(w.shape[2]) for w in weights], [int(w.shape[0]) for w in weights],
                          pad_mode, act, padding_idx, 0)
    L_ = _lib.lib()
    return bool(table.is_cuda and L_.rbr_textcnn_fwd_ws_bytes(C.byref(desc)) > 0 and L_.rbr_textcnn_bwd_prod_ws_bytes(C.byref(desc)) > 0)


_HEAD_NAMES = ("Wu", "bu", "Eu", "Wi", "bi", "Ei", "h", "g", "ub", "ib")
_HEAD_ACC = ("Eu", "Ei", "ub", "ib")       # gradients accumulated with atomics (rows addressed by id)


class _PairHead(torch.autograd.Function):
    """pred[B] = FM(LastFeat_u(u_feat, u_id), LastFeat_i(i_feat, i_id))  -- rbr_pair_head_* in rbr_hip.h."""

    @staticmethod
    def forward(ctx, u_feat, i_feat, u_id, i_id, drop, pad_u, pad_i, Wu, bu, Eu, Wi, bi, Ei, h, g, ub, ib):
        # i_feat None: u_feat is the [2B,H] output of the shared encoder, user rows first -- one tensor in, one gradient
        # tensor out (slicing it outside costs autograd two zero fills, two copies and an add per step)
        ctx.stacked = i_feat is None
        if ctx.stacked:
            pair = u_feat.contiguous()
            if pair.shape[0] % 2:
                raise RuntimeError(f"stacked pair features need an even row count, got {tuple(pair.shape)}")
            u_feat, i_feat = pair[:pair.shape[0] // 2], pair[pair.shape[0] // 2:]
        B, H = u_feat.shape
        K = Wu.shape[1]
        dev = u_feat.device
        L_ = _lib.lib()
        u_feat, i_feat = u_feat.contiguous(), i_feat.contiguous()
        u_id, i_id = u_id.contiguous(), i_id.contiguous()
        params = [t.contiguous() for t in (Wu, bu, Eu, Wi, bi, Ei, h, g, ub, ib)]
        hp = _lib.HeadParams(*[dev_ptr(t, F32, n) for t, n in zip(params, ("Wu", "bu", "Eu", "Wi", "bi", "Ei", "h", "g", "ub", "ib"))])
        ul = torch.empty(B, K, dtype=F32, device=dev)
        il = torch.empty(B, K, dtype=F32, device=dev)
        pred = torch.empty(B, dtype=F32, device=dev)
        ctx.flat = None
        if isinstance(drop, float):
            # training forward in one launch: the kernel draws the dropout multiplier itself (same stream of draws as
            # dropout_multiplier) and its spare workgroups clear the buffer the backward accumulates the embedding-style
            # gradients into
            p_drop = drop
            if p_drop >= 1.0:
                raise RuntimeError("pair_head: drop probability must be < 1 on the fused path")
            acc_n = sum(t.numel() for t, n in zip(params, _HEAD_NAMES) if n in _HEAD_ACC) if any(ctx.needs_input_grad[7:]) else 0
            flat = torch.empty(acc_n, dtype=F32, device=dev) if acc_n else None
            seed, state = _drop_rng(dev)
            drop = torch.empty(B, K, dtype=F32, device=dev) if p_drop > 0.0 else None
            check(L_.rbr_pair_head_fwd_train(B, H, K, dev_ptr(u_feat, F32, "u_feat"), dev_ptr(i_feat, F32, "i_feat"),
                                             dev_ptr(u_id, I64, "u_id"), dev_ptr(i_id, I64, "i_id"), C.byref(hp), float(p_drop),
                                             seed, state.data_ptr(), dev_ptr(drop, F32, "drop"), dev_ptr(flat, F32, "zero_buf"),
                                             acc_n, dev_ptr(ul, F32, "ul"), dev_ptr(il, F32, "il"), dev_ptr(pred, F32, "pred"),
                                             current_stream()), "rbr_pair_head_fwd_train")
            ctx.flat = flat
        else:
            if drop is not None:
                drop = drop.contiguous()
            check(L_.rbr_pair_head_fwd(B, H, K, dev_ptr(u_feat, F32, "u_feat"), dev_ptr(i_feat, F32, "i_feat"),
                                       dev_ptr(u_id, I64, "u_id"), dev_ptr(i_id, I64, "i_id"), C.byref(hp),
                                       dev_ptr(drop, F32, "drop"), dev_ptr(ul, F32, "ul"), dev_ptr(il, F32, "il"),
                                       dev_ptr(pred, F32, "pred"), current_stream()), "rbr_pair_head_fwd")
        ctx.dims = (B, H, K, int(pad_u), int(pad_i))
        ctx.has_drop = drop is not None
        ctx.save_for_backward(u_feat, i_feat, u_id, i_id, ul, il, *params, *([drop] if drop is not None else []))
        return pred

    @staticmethod
    def backward(ctx, d_pred):
        B, H, K, pad_u, pad_i = ctx.dims
        saved = list(ctx.saved_tensors)
        u_feat, i_feat, u_id, i_id, ul, il = saved[:6]
        params = saved[6:16]
        drop = saved[16] if ctx.has_drop else None
        dev = u_feat.device
        L_ = _lib.lib()
        names = ("Wu", "bu", "Eu", "Wi", "bi", "Ei", "h", "g", "ub", "ib")
        hp = _lib.HeadParams(*[dev_ptr(t, F32, n) for t, n in zip(params, names)])
        # embedding-style grads are accumulated with atomics -> start from zero (one fill for the four of them);
        # the rest is overwritten
        acc = [t for t, n in zip(params, names) if n in _HEAD_ACC]
        flat, ctx.flat = ctx.flat, None         # cleared by the training forward; a second backward gets a fresh one
        if flat is None:
            flat = torch.zeros(sum(t.numel() for t in acc), dtype=F32, device=dev)
        zeroed = iter(v.view_as(t) for v, t in zip(flat.split([t.numel() for t in acc]), acc))
        grads = [next(zeroed) if n in ("Eu", "Ei", "ub", "ib") else torch.empty_like(t) for t, n in zip(params, names)]
        hg = _lib.HeadGrads(*[dev_ptr(t, F32, "d" + n) for t, n in zip(grads, names)])
        d_pair = torch.empty(2 * B, H, dtype=F32, device=dev)
        d_uf, d_if = d_pair[:B], d_pair[B:]
        wsn = L_.rbr_pair_head_bwd_ws_floats(B, K)
        ws = torch.empty(wsn, dtype=F32, device=dev) if wsn else None
        d_pred = d_pred.contiguous()
        check(L_.rbr_pair_head_bwd(B, H, K, dev_ptr(u_feat, F32, "u_feat"), dev_ptr(i_feat, F32, "i_feat"),
                                   dev_ptr(u_id, I64, "u_id"), dev_ptr(i_id, I64, "i_id"), C.byref(hp),
                                   dev_ptr(drop, F32, "drop"), dev_ptr(ul, F32, "ul"), dev_ptr(il, F32, "il"),
                                   dev_ptr(d_pred, F32, "d_pred"), pad_u, pad_i, C.byref(hg),
                                   dev_ptr(d_uf, F32, "d_ufeat"), dev_ptr(d_if, F32, "d_ifeat"), dev_ptr(ws, F32, "ws"),
                                   current_stream()), "rbr_pair_head_bwd")
        if ctx.stacked:
            return (d_pair, None, None, None, None, None, None, *grads)
        return (d_uf, d_if, None, None, None, None, None, *grads)


def pair_head(u_feat, i_feat, u_id, i_id, Wu, bu, Eu, Wi, bi, Ei, h, g, ub, ib, *, drop=None, pad_u=0, pad_i=0):
    """LastFeat x2 + FM.  u_feat/i_feat [B,H] (or u_feat [2B,H] = user rows then item rows, i_feat None);
    ids [B] int64; returns pred [B].  drop: None, a [B,K] multiplier tensor, or a float = dropout probability of a
    TRAINING forward (the kernel draws the multiplier itself: one launch less than dropout_multiplier + pair_head)."""
    return _PairHead.apply(u_feat, i_feat, u_id, i_id, drop, pad_u, pad_i, Wu, bu, Eu, Wi, bi, Ei, h, g, ub, ib)


# --------------------------------------------------------------------------- encoder + head of a two-tower model, fused
class _LossRequest:
    """fused_loss(target): the trainer's nn.MSELoss target, announced to the model's forward so that a fused head launch can
    compute the loss as well; loss_for(pred) hands it out (None: the forward did not take the offer)."""

    def __init__(self, target):
        self.target, self.pred_ptr, self.loss = target, None, None

    def loss_for(self, pred: torch.Tensor):
        if self.loss is not None and self.pred_ptr == pred.data_ptr() and pred.numel() == self.target.numel():
            return self.loss
        return None


_LOSS_REQUEST: list = []


@contextlib.contextmanager
def fused_loss(target: torch.Tensor):
    """with fused_loss(ratings) as fl: pred = model(*batch); loss = fl.loss_for(pred) -- the mean-squared error against
    `target` (trainer/train_deepconn_pp.py:164) out of the model's last launch when the model supports it (DeepCoNN's fused
    encoder + head), else None and the caller computes mse_loss(pred, target) itself."""
    req = _LossRequest(target)
    _LOSS_REQUEST.append(req)
    try:
        yield req
    finally:
        _LOSS_REQUEST.remove(req)


_TICKETS: dict = {}


def _ticket(dev) -> torch.Tensor:
    """Arrival counter of rbr_pair_head_fwd_pool's MSE reduction (the last pair block to arrive sums the loss and puts the counter
    back to 0): one per (device, STREAM) -- launches on one stream are ordered, so they may share it; two fused heads in flight
    on different streams (a side-stream eval beside a training step, a graph replay beside an eager step) must not."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(4, dtype=I32, device=dev)
    return t


def prepare_stream_state(dev) -> None:
    """Creates the per-(device, stream) state of the CURRENT stream -- the arrival counter of the fused head's MSE reduction --
    so that its first use does not allocate and zero-fill it.  train_step.GraphedTrainStep /
    GraphedForward call it on their capture stream before recording: a counter buffer created during a capture would live in
    that graph's private memory pool and cost every replay a fill node."""
    dev = torch.device(dev)
    _ticket(dev)


def encode_head_applicable(table, n_docs, L, kernel_sizes, channels, padding_idx) -> bool:
    """True when the fused encoder + head function serves this conv: the token-product formulation applies and one
    pool-epilogue launch covers every channel slot."""
    if not table.is_cuda or table.dtype != F32 or n_docs % 2 or os.environ.get("RBR_FUSED_STEP", "1") == "0":
        return False
    ntiles, wpd = (sum(channels) + 31) // 32, (L + 31) // 32
    if ntiles > 8 or (2 * sum(channels) + 2 * wpd * ntiles * 32 + 2 * wpd) * 4 > 60 * 1024:      # rbr_pair_head_fwd_pool's limits
        return False
    desc = _lib.make_desc(n_docs, L, table.shape[1], table.shape[0], kernel_sizes, channels, PAD_SAME, ACT_RELU, padding_idx)
    return _lib.lib().rbr_textcnn_fwd_ws_bytes(C.byref(desc)) > 0


class _EncodeHead(torch.autograd.Function):
    """pred[B] (and the MSE loss against `target`) = rating head(TextCNN(user docs), TextCNN(item docs), ids): the training /
    eval forward of DeepCoNN++ (deepconn.py:43-53) as 6 launches -- id check + list state | token marks + slab scan | token list
    + weight images | distinct-token GEMM | gather + max-pool (+ clearing the backward's G) | pool epilogue + LastFeat x2 + FM
    (+ MSELoss) -- and its backward as head_bwd, then [zero G, G build, G @ Wprod^T] beside [dW] (see _textcnn_backward)."""

    @staticmethod
    def forward(ctx, table, id_sets, ids, mask, u_id, i_id, drop, target, padding_idx, pad_u, pad_i, n_widths, first, *params):
        ws_, bs_ = [w.contiguous() for w in params[:n_widths]], [b.contiguous() for b in params[n_widths:2 * n_widths]]
        head = [t.contiguous() for t in params[2 * n_widths:]]
        L_ = _lib.lib()
        dev = table.device
        table_c = table.contiguous()
        dev_ptr(table_c, F32, "word table")
        V, D = table.shape
        kernel_sizes = [int(w.shape[2]) for w in ws_]
        channels = [int(w.shape[0]) for w in ws_]
        Ctot = sum(channels)
        st = current_stream()
        # ---- clean id tensors: token ids of both towers first (the conv's [2B, L] batch), then the rest
        if id_sets is not None:
            tens = [t.contiguous() for t, _, _ in id_sets]
            flat = torch.empty(sum(t.numel() for t in tens), dtype=I64, device=dev)
            arr = (_lib.IdSet * len(tens))()
            outs, o = [], 0
            for k, (t, (_, rows, rep)) in enumerate(zip(tens, id_sets)):
                dev_ptr(t, I64, "ids")
                v = flat[o:o + t.numel()].view(t.shape)
                outs.append(v)
                o += t.numel()
                arr[k] = _lib.IdSet(t.data_ptr(), v.data_ptr(), t.numel(), int(rows), 0 if rep is None or rep < 0 else int(rep))
            n_tok = tens[0].numel() + tens[1].numel()
            ids = flat[:n_tok].view(2 * tens[0].shape[0], tens[0].shape[1])
            u_id, i_id = outs[2], outs[3]
        else:
            ids, u_id, i_id = ids.contiguous(), u_id.contiguous(), i_id.contiguous()
        n_docs, L = ids.shape
        B = n_docs // 2
        mask8 = _mask_u8(mask)
        if mask8 is not None and mask8.shape != ids.shape:
            raise AssertionError("inputs.shape[:-1] == masks.shape")   # deepconn/utils.py:58
        desc = _lib.make_desc(n_docs, L, D, V, kernel_sizes, channels, PAD_SAME, ACT_RELU, padding_idx)
        n_part = L_.rbr_textcnn_partial_elems(C.byref(desc))
        ws_bytes = L_.rbr_textcnn_fwd_ws_bytes(C.byref(desc))
        if not n_part or not ws_bytes:
            check(-1, "encode_head plan (encode_head_applicable() was not consulted)")
        pval = torch.empty(n_part, dtype=F32, device=dev)
        pidx = torch.empty(n_part, dtype=I32, device=dev)
        feat = torch.empty(n_docs, Ctot, dtype=F32, device=dev)
        argmax = torch.empty(n_docs, Ctot, dtype=I32, device=dev)
        prod_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        wsp = prod_ws.data_ptr()
        training = any(ctx.needs_input_grad)          # (grad mode itself is off inside forward)
        ev = TIMER.record("textcnn_prod_prepare")
        if id_sets is not None:
            check(L_.rbr_textcnn_prod_prepare_ids(C.byref(desc), len(tens), arr, _id_err(dev).data_ptr(), dev_ptr(ids, I64, "ids"),
                                                  dev_ptr(mask8, U8, "mask"), ptr_array(ws_, F32, "conv weight"),
                                                  dev_ptr(pidx, I32, "pidx"), wsp, st), "rbr_textcnn_prod_prepare_ids")
        else:
            check(L_.rbr_textcnn_prod_prepare(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                              ptr_array(ws_, F32, "conv weight"), dev_ptr(pidx, I32, "pidx"), wsp, st),
                  "rbr_textcnn_prod_prepare")
        if ev is not None:
            ev.record()
        ev = TIMER.record("textcnn_prod_table")
        check(L_.rbr_textcnn_prod_table(C.byref(desc), dev_ptr(table_c, F32, "word table"), wsp, st), "rbr_textcnn_prod_table")
        if ev is not None:
            ev.record()
        ev = TIMER.record("textcnn_prod_pool")
        check(L_.rbr_textcnn_prod_pool(C.byref(desc), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"), None,
                                       dev_ptr(pval, F32, "pval"), dev_ptr(pidx, I32, "pidx"), wsp, st), "rbr_textcnn_prod_pool")
        if ev is not None:
            ev.record()
        # ---- pool epilogue + rating head (+ loss)
        K = head[0].shape[1]
        hp = _lib.HeadParams(*[dev_ptr(t, F32, n) for t, n in zip(head, _HEAD_NAMES)])
        ul = torch.empty(B, K, dtype=F32, device=dev)
        il = torch.empty(B, K, dtype=F32, device=dev)
        pred = torch.empty(B, dtype=F32, device=dev)
        p_drop, drop_t, flat_zero, acc_n = 0.0, None, None, 0
        seed, state = 0, None
        if isinstance(drop, float):
            if drop >= 1.0:
                raise RuntimeError("pair_head: drop probability must be < 1 on the fused path")
            p_drop = drop
            acc_n = sum(t.numel() for t, n in zip(head, _HEAD_NAMES) if n in _HEAD_ACC) if training else 0
            flat_zero = torch.empty(acc_n, dtype=F32, device=dev) if acc_n else None
            if p_drop > 0.0:
                seed, state = _drop_rng(dev)
                drop_t = torch.empty(B, K, dtype=F32, device=dev)
        elif drop is not None:
            drop_t = drop.contiguous()
        loss = d_unit = None
        if target is not None:
            target = target.contiguous()
            loss = torch.empty((), dtype=F32, device=dev)
            d_unit = torch.empty(B, dtype=F32, device=dev)
        ev = TIMER.record("pair_head_fwd_pool")
        check(L_.rbr_pair_head_fwd_pool(C.byref(desc), dev_ptr(pval, F32, "pval"), dev_ptr(pidx, I32, "pidx"),
                                        ptr_array(bs_, F32, "conv bias"), dev_ptr(feat, F32, "feat"), dev_ptr(argmax, I32, "argmax"),
                                        dev_ptr(first, I64, "first"), K, dev_ptr(u_id, I64, "u_id"), dev_ptr(i_id, I64, "i_id"), C.byref(hp),
                                        dev_ptr(drop_t, F32, "drop") if p_drop == 0.0 else None, float(p_drop), seed,
                                        state.data_ptr() if state is not None else None,
                                        dev_ptr(drop_t, F32, "drop") if p_drop > 0.0 else None, dev_ptr(flat_zero, F32, "zero_buf"),
                                        acc_n, dev_ptr(ul, F32, "ul"), dev_ptr(il, F32, "il"), dev_ptr(pred, F32, "pred"),
                                        dev_ptr(target, F32, "target"), dev_ptr(loss, F32, "loss"), dev_ptr(d_unit, F32, "d_unit"),
                                        _ticket(dev).data_ptr() if target is not None else None, st), "rbr_pair_head_fwd_pool")
        if ev is not None:
            ev.record()
        ctx.conv = _ConvSaved(table=table_c, ids=ids, packed=None, feat=feat, argmax=argmax, mask8=mask8, gate=None, ws=ws_,
                              desc=desc, prod_ws=prod_ws, fanout_acc=None, bws=None)
        ctx.head = (u_id, i_id, ul, il, head, drop_t, flat_zero, d_unit)
        ctx.first = first
        ctx.dims = (B, Ctot, K, int(pad_u), int(pad_i), n_widths)
        ctx.set_materialize_grads(False)
        if loss is None:
            loss = torch.empty((), dtype=F32, device=dev)      # no target: an unwritten placeholder nobody reads
            ctx.mark_non_differentiable(loss)
        return pred, loss

    @staticmethod
    def backward(ctx, d_pred, d_loss):
        B, H, K, pad_u, pad_i, n_widths = ctx.dims
        u_id, i_id, ul, il, head, drop_t, flat, d_unit = ctx.head
        S = ctx.conv
        dev = S.table.device
        L_ = _lib.lib()
        if d_loss is not None:                  # the fused loss: d loss / d pred was written by the forward launch
            unit = _UNIT.get(dev)
            g = d_unit if (unit is not None and d_loss.data_ptr() == unit.data_ptr()) else d_unit * d_loss
            d_pred = g if d_pred is None else d_pred + g
        if d_pred is None:
            d_pred = torch.zeros(B, dtype=F32, device=dev)
        d_pred = d_pred.contiguous()
        hp = _lib.HeadParams(*[dev_ptr(t, F32, n) for t, n in zip(head, _HEAD_NAMES)])
        accs = [t for t, n in zip(head, _HEAD_NAMES) if n in _HEAD_ACC]
        if flat is None:
            flat = torch.zeros(sum(t.numel() for t in accs), dtype=F32, device=dev)
        ctx.head = (u_id, i_id, ul, il, head, drop_t, None, d_unit)      # a second backward gets a fresh accumulation buffer
        zeroed = iter(v.view_as(t) for v, t in zip(flat.split([t.numel() for t in accs]), accs))
        grads = [next(zeroed) if n in _HEAD_ACC else torch.empty_like(t) for t, n in zip(head, _HEAD_NAMES)]
        hg = _lib.HeadGrads(*[dev_ptr(t, F32, "d" + n) for t, n in zip(grads, _HEAD_NAMES)])
        d_pair = torch.empty(2 * B, H, dtype=F32, device=dev)
        feat = S.feat
        need_conv = any(ctx.needs_input_grad[13:13 + 2 * n_widths]) or ctx.needs_input_grad[0]
        check(L_.rbr_pair_head_bwd(B, H, K, dev_ptr(feat[:B], F32, "u_feat"), dev_ptr(feat[B:], F32, "i_feat"),
                                   dev_ptr(u_id, I64, "u_id"), dev_ptr(i_id, I64, "i_id"), C.byref(hp),
                                   dev_ptr(drop_t, F32, "drop"), dev_ptr(ul, F32, "ul"), dev_ptr(il, F32, "il"),
                                   dev_ptr(d_pred, F32, "d_pred"), pad_u, pad_i, C.byref(hg),
                                   dev_ptr(d_pair[:B], F32, "d_ufeat"), dev_ptr(d_pair[B:], F32, "d_ifeat"), None,
                                   current_stream()), "rbr_pair_head_bwd")
        if need_conv and ctx.first is not None:      # in-batch dedup: the repeated documents' gradient rows onto their first occurrence
            check(L_.rbr_dedup_fold_rows(2 * B, H, dev_ptr(ctx.first, I64, "first"), dev_ptr(d_pair, F32, "d_feat"), current_stream()),
                  "rbr_dedup_fold_rows")
        if need_conv:
            dtable, _, dWs, dbs = _textcnn_backward(S, d_pair, ctx.needs_input_grad[0], False)
        else:
            dtable, dWs, dbs = None, [None] * n_widths, [None] * n_widths
        return (dtable, None, None, None, None, None, None, None, None, None, None, None, None, *dWs, *dbs, *grads)


def encode_head(table, ids, mask, u_id, i_id, conv_weights, conv_biases, head_params, *, id_sets=None, drop=None,
                padding_idx=0, pad_u=0, pad_i=0, first=None):
    """DeepCoNN++'s forward from token ids to predictions in one autograd function (see _EncodeHead).  ids / mask: the stacked
    [2B, L] towers (user rows first) -- or id_sets = [(u_docs, V, pad), (i_docs, V, pad), (u_ids, U, 0), (i_ids, I, 0)], the raw
    id tensors, whose range check then rides in the prepare stage's first launch (ids / u_id / i_id are ignored).
    head_params: (Wu, bu, Eu, Wi, bi, Ei, h, g, ub, ib).  drop: None, a [B, K] multiplier, or the dropout probability of a
    training forward.  first [2B] (dedup_rows; `mask` then carries the blanked rows): document r's features are document
    first[r]'s -- the in-batch dedup by id, inside the fused step.  Inside `with fused_loss(target)` the MSE against target is computed by the head launch as well."""
    req = _LOSS_REQUEST[-1] if _LOSS_REQUEST else None
    target = None
    if req is not None and req.loss is None and req.target.is_cuda and req.target.dtype == F32 and req.target.dim() == 1 \
            and torch.is_grad_enabled():
        B = (id_sets[0][0].shape[0] if id_sets is not None else ids.shape[0] // 2)
        if req.target.shape[0] == B:
            target = req.target
    pred, loss = _EncodeHead.apply(table, id_sets, ids, mask, u_id, i_id, drop, target, padding_idx, pad_u, pad_i,
                                   len(conv_weights), first, *conv_weights, *conv_biases, *head_params)
    if target is not None:
        req.pred_ptr, req.loss = pred.data_ptr(), loss
    return pred


def stack_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], 0) of two input tensors; no copy when b already follows a in the same allocation (the
    static batch buffers of train_step.GraphedTrainStep, or a loader that delivers both towers in one block)."""
    if (a.dtype == b.dtype and a.shape[1:] == b.shape[1:] and a.is_contiguous() and b.is_contiguous()
            and not a.requires_grad and not b.requires_grad and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        return a.as_strided((a.shape[0] + b.shape[0], *a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat([a, b], dim=0)


def clone_adjacent(tensors):
    """Clones of `tensors`; neighbours (2k, 2k+1) of equal shape and dtype share one allocation, first then second,
    so stack_rows() on the clones is a view."""
    out = list(tensors)
    k = 0
    while k < len(out):
        a = out[k]
        b = out[k + 1] if k + 1 < len(out) else None
        if b is not None and a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and a.dim() >= 1:
            both = torch.empty((2 * a.shape[0], *a.shape[1:]), dtype=a.dtype, device=a.device)
            both[:a.shape[0]].copy_(a)
            both[a.shape[0]:].copy_(b)
            out[k], out[k + 1] = both[:a.shape[0]], both[a.shape[0]:]
            k += 2
        else:
            out[k] = a.clone()
            k += 1
    return tuple(out)


def dropout_multiplier(shape, p: float, training: bool, device, lane: int = 0) -> Optional[torch.Tensor]:
    """The multiplier F.dropout would apply (0 or 1/(1-p)): rbr_dropout_multiplier, keyed by the device generator's seed
    (torch.manual_seed) and a per-device call counter that lives on the device."""
    if not training or p <= 0.0:
        return None
    if p >= 1.0:
        return torch.zeros(shape, dtype=F32, device=device)
    # own Philox kernel with the call counter in device memory: one launch, and -- unlike a torch RNG op -- nothing for the
    # host to patch before every replay of a captured step (the generator's seed / offset fills)
    dev = torch.device(device)
    seed, state = _drop_rng(dev, lane)
    out = torch.empty(shape, dtype=F32, device=dev)
    check(_lib.lib().rbr_dropout_multiplier(out.numel(), float(p), seed, state.data_ptr(), dev_ptr(out, F32, "out"),
                                            current_stream()), "rbr_dropout_multiplier")
    return out


def _drop_rng(dev, lane: int = 0):
    """(seed, device state) of the dropout draws on `dev`: the device generator's seed (torch.manual_seed) and the
    [call number, ticket] pair that lives on the device.  `lane` selects an independent stream of draws: launches that may
    run concurrently (the second tower on its own HIP stream, second_tower()) must not share a call counter."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise RuntimeError(f"dropout: device must be a HIP device (got {dev}); there is no CPU path")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    state = _DROP_STATE.get((idx, lane))
    if state is None:
        state = _DROP_STATE[(idx, lane)] = torch.zeros(2, dtype=torch.int64, device=dev)
    seed = (torch.cuda.default_generators[idx].initial_seed() + 0x9E3779B97F4A7C15 * lane) & 0xFFFFFFFFFFFFFFFF
    return seed, state


_DROP_STATE: dict = {}       # per device: [call number, workgroup ticket] (uint64 x 2), advanced on the device



class _MseLoss(torch.autograd.Function):
    """mean((pred - target)^2) -- the trainers' nn.MSELoss (train_deepconn_pp.py:137,164).  The forward launch also writes
    d loss / d pred for an upstream gradient of 1; a backward that is handed unit_scalar() (train_step does that) returns it
    without a launch, any other upstream gradient costs one launch."""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.contiguous()
        if pred.shape != target.shape:
            raise RuntimeError(f"mse_loss: pred {tuple(pred.shape)} vs target {tuple(target.shape)}")
        loss = torch.empty((), dtype=F32, device=pred.device)
        d_unit = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        check(_lib.lib().rbr_mse_loss_fwd(pred.numel(), dev_ptr(pred, F32, "pred"), dev_ptr(target, F32, "target"),
                                          dev_ptr(loss, F32, "loss"), dev_ptr(d_unit, F32, "d_pred_unit"), current_stream()),
              "rbr_mse_loss_fwd")
        ctx.save_for_backward(pred, target, *([d_unit] if d_unit is not None else []))
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        pred, target = ctx.saved_tensors[:2]
        unit = _UNIT.get(pred.device)
        if unit is not None and len(ctx.saved_tensors) == 3 and d_loss.data_ptr() == unit.data_ptr():
            return ctx.saved_tensors[2], None
        d_pred = torch.empty_like(pred)
        d_loss = d_loss.contiguous()
        check(_lib.lib().rbr_mse_loss_bwd(pred.numel(), dev_ptr(pred, F32, "pred"), dev_ptr(target, F32, "target"),
                                          dev_ptr(d_loss, F32, "d_loss"), dev_ptr(d_pred, F32, "d_pred"), current_stream()),
              "rbr_mse_loss_bwd")
        return d_pred, None


_UNIT: dict = {}


def unit_scalar(device) -> torch.Tensor:
    """The constant 1.0 on `device` (one tensor per device, never written): pass it to loss.backward() as the root gradient
    to spare the per-step fill, and mse_loss's backward recognises it."""
    device = torch.device(device)
    t = _UNIT.get(device)
    if t is None:
        t = _UNIT[device] = torch.ones((), dtype=F32, device=device)
    return t


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """F.mse_loss(pred, target) (mean reduction) for f32 device tensors of equal shape."""
    return _MseLoss.apply(pred, target)


# --------------------------------------------------------------------------- NARRE attention pool
class _ReviewAttn(torch.autograd.Function):
    """out[B,H], att[B,R,1] = LinearAttention(feat[B,R,H], other_id[B,R])  -- rbr_review_attn_* in rbr_hip.h."""

    @staticmethod
    def forward(ctx, feat, other_id, pad_idx, W_rv, W_id, h, b1, b2, ebd, drop=None):
        B, R, H = feat.shape
        A = W_rv.shape[1]
        dev = feat.device
        L_ = _lib.lib()
        feat = feat.contiguous()
        other_id = other_id.contiguous()
        params = [t.contiguous() for t in (W_rv, W_id, h, b1, b2, ebd)]
        names = ("W_rv", "W_id", "h", "b1", "b2", "ebd")
        ap = _lib.AttnParams(*[dev_ptr(t, F32, n) for t, n in zip(params, names)])
        out = torch.empty(B, H, dtype=F32, device=dev)
        att = torch.empty(B, R, 1, dtype=F32, device=dev)
        hid = torch.empty(B, R, A, dtype=F32, device=dev)
        drop = drop.contiguous() if drop is not None else None
        if drop is not None and tuple(drop.shape) != (B, H):
            raise RuntimeError(f"review_attention: dropout multiplier {tuple(drop.shape)} is not [B, H] = {(B, H)}")
        check(L_.rbr_review_attn_fwd(B, R, H, A, dev_ptr(feat, F32, "feat"), dev_ptr(other_id, I64, "other_id"), C.byref(ap),
                                     dev_ptr(drop, F32, "drop"), dev_ptr(out, F32, "out"), dev_ptr(att, F32, "att"),
                                     dev_ptr(hid, F32, "hid"), current_stream()), "rbr_review_attn_fwd")
        ctx.dims = (B, R, H, A, int(pad_idx))
        ctx.has_drop = drop is not None
        ctx.set_materialize_grads(False)     # an unused `att` output arrives as None in backward, not as a freshly filled zero tensor
        ctx.save_for_backward(feat, other_id, att, hid, *params, *([drop] if drop is not None else []))
        return out, att

    @staticmethod
    def backward(ctx, d_out, d_att):
        B, R, H, A, pad_idx = ctx.dims
        feat, other_id, att, hid = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:10]
        drop = ctx.saved_tensors[10] if ctx.has_drop else None
        names = ("W_rv", "W_id", "h", "b1", "b2", "ebd")
        dev = feat.device
        L_ = _lib.lib()
        ap = _lib.AttnParams(*[dev_ptr(t, F32, n) for t, n in zip(params, names)])
        grads = [torch.zeros_like(t) if n == "ebd" else torch.empty_like(t) for t, n in zip(params, names)]
        ag = _lib.AttnGrads(*[dev_ptr(t, F32, "d" + n) for t, n in zip(grads, names)])
        d_feat = torch.empty_like(feat)
        ws = torch.empty(L_.rbr_review_attn_bwd_ws_floats(B, R, H, A), dtype=F32, device=dev)
        d_out = d_out.contiguous() if d_out is not None else torch.zeros(B, H, dtype=F32, device=dev)
        d_att = d_att.contiguous() if d_att is not None else None
        check(L_.rbr_review_attn_bwd(B, R, H, A, dev_ptr(feat, F32, "feat"), dev_ptr(other_id, I64, "other_id"), C.byref(ap),
                                     dev_ptr(drop, F32, "drop"), dev_ptr(att, F32, "att"), dev_ptr(hid, F32, "hid"),
                                     dev_ptr(d_out, F32, "d_out"), dev_ptr(d_att, F32, "d_att"), pad_idx, C.byref(ag),
                                     dev_ptr(d_feat, F32, "d_feat"), dev_ptr(ws, F32, "ws"), current_stream()),
              "rbr_review_attn_bwd")
        return (d_feat, None, None, *grads, None)


class _ReviewAttn2(torch.autograd.Function):
    """Both LinearAttention pools of a two-tower model in one launch per stage -- rbr_review_attn2_* in rbr_hip.h.
    feat [2,B,R,H], other_id [2,B,R], drop [2,B,H] or None, then the six parameters of side 0 and of side 1."""

    NAMES = ("W_rv", "W_id", "h", "b1", "b2", "ebd")

    @staticmethod
    def forward(ctx, feat, other_id, pad0, pad1, drop, *params):
        _, B, R, H = feat.shape
        A = params[0].shape[1]
        dev = feat.device
        L_ = _lib.lib()
        feat, other_id = feat.contiguous(), other_id.contiguous()
        params = [t.contiguous() for t in params]
        names = _ReviewAttn2.NAMES
        aps = [_lib.AttnParams(*[dev_ptr(t, F32, n) for t, n in zip(params[6 * k:6 * k + 6], names)]) for k in range(2)]
        drop = drop.contiguous() if drop is not None else None
        if drop is not None and tuple(drop.shape) != (2, B, H):
            raise RuntimeError(f"review_attention2: dropout multiplier {tuple(drop.shape)} is not [2, B, H] = {(2, B, H)}")
        out = torch.empty(2, B, H, dtype=F32, device=dev)
        att = torch.empty(2, B, R, 1, dtype=F32, device=dev)
        hid = torch.empty(2, B, R, A, dtype=F32, device=dev)
        check(L_.rbr_review_attn2_fwd(B, R, H, A, dev_ptr(feat, F32, "feat"), dev_ptr(other_id, I64, "other_id"), C.byref(aps[0]),
                                      C.byref(aps[1]), dev_ptr(drop, F32, "drop"), dev_ptr(out, F32, "out"), dev_ptr(att, F32, "att"),
                                      dev_ptr(hid, F32, "hid"), current_stream()), "rbr_review_attn2_fwd")
        ctx.dims = (B, R, H, A, int(pad0), int(pad1))
        ctx.has_drop = drop is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(feat, other_id, att, hid, *params, *([drop] if drop is not None else []))
        return out, att

    @staticmethod
    def backward(ctx, d_out, d_att):
        B, R, H, A, pad0, pad1 = ctx.dims
        feat, other_id, att, hid = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:16]
        drop = ctx.saved_tensors[16] if ctx.has_drop else None
        names = _ReviewAttn2.NAMES
        dev = feat.device
        L_ = _lib.lib()
        aps = [_lib.AttnParams(*[dev_ptr(t, F32, n) for t, n in zip(params[6 * k:6 * k + 6], names)]) for k in range(2)]
        grads = [torch.empty_like(t) for t in params]           # the embedding-row gradients are zeroed by the call
        ags = [_lib.AttnGrads(*[dev_ptr(t, F32, "d" + n) for t, n in zip(grads[6 * k:6 * k + 6], names)]) for k in range(2)]
        d_feat = torch.empty_like(feat)
        ws = torch.empty(2 * L_.rbr_review_attn_bwd_ws_floats(B, R, H, A), dtype=F32, device=dev)
        d_out = d_out.contiguous() if d_out is not None else torch.zeros(2, B, H, dtype=F32, device=dev)
        d_att = d_att.contiguous() if d_att is not None else None
        check(L_.rbr_review_attn2_bwd(B, R, H, A, dev_ptr(feat, F32, "feat"), dev_ptr(other_id, I64, "other_id"), C.byref(aps[0]),
                                      C.byref(aps[1]), dev_ptr(drop, F32, "drop"), dev_ptr(att, F32, "att"), dev_ptr(hid, F32, "hid"),
                                      dev_ptr(d_out, F32, "d_out"), dev_ptr(d_att, F32, "d_att"), pad0, pad1, C.byref(ags[0]),
                                      C.byref(ags[1]), params[5].shape[0], params[11].shape[0], dev_ptr(d_feat, F32, "d_feat"),
                                      dev_ptr(ws, F32, "ws"), current_stream()), "rbr_review_attn2_bwd")
        return (d_feat, None, None, None, None, *grads)


def review_attention2(feat, other_id, params0, params1, *, pad_idx=(0, 0), drop=None):
    """The user and the item LinearAttention of NARRE in one launch per stage: feat [2,B,R,H] (side 0 first), other_id [2,B,R],
    params0 / params1 = (W_rv, W_id, h, b1, b2, ebd) of the sides, drop [2,B,H] or None.  Returns (out [2,B,H], att [2,B,R,1])."""
    return _ReviewAttn2.apply(feat, other_id, pad_idx[0], pad_idx[1], drop, *params0, *params1)


def review_attention(feat, other_id, W_rv, W_id, h, b1, b2, ebd, *, pad_idx=0, drop=None):
    """NARRE LinearAttention: returns (out [B,H], att [B,R,1]).  `drop` [B,H]: the multiplier of the nn.Dropout that follows
    (dropout_multiplier), applied inside the kernels -- forward and backward -- instead of by two elementwise launches."""
    return _ReviewAttn.apply(feat, other_id, pad_idx, W_rv, W_id, h, b1, b2, ebd, drop)


class _BlockCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, d):
        a, b, c, d = (t.contiguous() for t in (a, b, c, d))
        B, C1, C2 = a.shape[0], a.shape[1], b.shape[1]
        if c.shape != a.shape or d.shape != b.shape or b.shape[0] != B:
            raise RuntimeError("block_cat: expected a, c [B, C1] and b, d [B, C2]")
        out = torch.empty(2 * B, C1 + C2, dtype=F32, device=a.device)
        check(_lib.lib().rbr_block_cat(B, C1, C2, dev_ptr(a, F32, "a"), dev_ptr(b, F32, "b"), dev_ptr(c, F32, "c"),
                                       dev_ptr(d, F32, "d"), dev_ptr(out, F32, "out"), current_stream()), "rbr_block_cat")
        ctx.dims = (B, C1, C2)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C1, C2 = ctx.dims
        g = g.contiguous()
        a, c = (torch.empty(B, C1, dtype=F32, device=g.device) for _ in range(2))
        b, d = (torch.empty(B, C2, dtype=F32, device=g.device) for _ in range(2))
        check(_lib.lib().rbr_block_split(B, C1, C2, dev_ptr(g, F32, "g"), dev_ptr(a, F32, "a"), dev_ptr(b, F32, "b"),
                                         dev_ptr(c, F32, "c"), dev_ptr(d, F32, "d"), current_stream()), "rbr_block_split")
        return a, b, c, d


def block_cat(a, b, c, d):
    """[[a, b], [c, d]] as one [2B, C1 + C2] tensor (a, c [B, C1]; b, d [B, C2]): torch.cat((cat((a, b), 1), cat((c, d), 1)), 0)
    in one launch, and contiguous gradient pieces from one launch in the backward."""
    return _BlockCat.apply(a, b, c, d)


class _PairDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, K = x.shape[0] // 2, x.shape[1]
        out = torch.empty(B, dtype=F32, device=x.device)
        check(_lib.lib().rbr_pair_dot_fwd(B, K, dev_ptr(x, F32, "x"), dev_ptr(out, F32, "out"), current_stream()), "rbr_pair_dot_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        B, K = x.shape[0] // 2, x.shape[1]
        d_x = torch.empty_like(x)
        d_out = d_out.contiguous()
        check(_lib.lib().rbr_pair_dot_bwd(B, K, dev_ptr(x, F32, "x"), dev_ptr(d_out, F32, "d_out"), dev_ptr(d_x, F32, "d_x"),
                                          current_stream()), "rbr_pair_dot_bwd")
        return d_x


def pair_dot(stacked: torch.Tensor) -> torch.Tensor:
    """ratings[b] = sum_k stacked[b, k] * stacked[B + b, k] for a [2B, K] block with the user rows first (D-ATT's
    `torch.sum(torch.mul(u_feat, i_feat), 1)`, dual_att.py:58, on the shared fc's stacked output)."""
    if stacked.dim() != 2 or stacked.shape[0] % 2:
        raise RuntimeError(f"pair_dot: expected a [2B, K] block, got {tuple(stacked.shape)}")
    return _PairDot.apply(stacked)


# --------------------------------------------------------------------------- nn.Linear on MFMA
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, relu, drop):
        N, IN = x.shape
        OUT = W.shape[0]
        L_ = _lib.lib()
        x, W = x.contiguous(), W.contiguous()
        b = b.contiguous() if b is not None else None
        drop = drop.contiguous() if drop is not None else None
        y = torch.empty(N, OUT, dtype=F32, device=x.device)
        wsn = L_.rbr_linear_fwd_ws_floats(N, IN, OUT)          # > 0: the product is split along K (few output tiles)
        ws = torch.empty(wsn, dtype=F32, device=x.device) if wsn else None
        check(L_.rbr_linear_fwd_ex(N, IN, OUT, dev_ptr(x, F32, "x"), dev_ptr(W, F32, "W"), dev_ptr(b, F32, "b"), int(relu),
                                   dev_ptr(drop, F32, "drop"), dev_ptr(y, F32, "y"), dev_ptr(ws, F32, "ws"),
                                   current_stream()), "rbr_linear_fwd_ex")
        ctx.relu = int(relu)
        ctx.has_b = b is not None
        ctx.has_drop = drop is not None
        ctx.save_for_backward(x, W, y, *([drop] if drop is not None else []))
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, W, y = ctx.saved_tensors[:3]
        drop = ctx.saved_tensors[3] if ctx.has_drop else None
        N, IN = x.shape
        OUT = W.shape[0]
        L_ = _lib.lib()
        dev = x.device
        d_y = d_y.contiguous()
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        db = torch.empty(OUT, dtype=F32, device=dev) if ctx.has_b else None
        ws = torch.empty(L_.rbr_linear_bwd_ex_ws_floats(N, IN, OUT), dtype=F32, device=dev)
        check(L_.rbr_linear_bwd_ex(N, IN, OUT, dev_ptr(x, F32, "x"), dev_ptr(W, F32, "W"), dev_ptr(y, F32, "y"),
                                   dev_ptr(d_y, F32, "d_y"), ctx.relu, dev_ptr(drop, F32, "drop"), dev_ptr(d_x, F32, "d_x"),
                                   dev_ptr(dW, F32, "dW"), dev_ptr(db, F32, "db"), dev_ptr(ws, F32, "ws"),
                                   current_stream()), "rbr_linear_bwd_ex")
        return d_x, dW, db, None, None


def linear(x, W, b=None, *, relu=False, drop=None, tanh=False):
    """y = act(x @ W^T + b) * drop  with torch's nn.Linear weight layout [OUT, IN]; act = ReLU, Tanh or none."""
    return _Linear.apply(x, W, b, 2 if tanh else int(bool(relu)), drop)


class _ConvShiftAdd(torch.autograd.Function):
    """out [bz, C, L] = bias + the kz shifted adds of the product T [bz * L, sum kz*ch]  -- rbr_conv_shift_add_* in rbr_hip.h."""

    @staticmethod
    def forward(ctx, t, bz, L, kernel_sizes, channels, *biases):
        t = t.contiguous()
        bs = [b.contiguous() for b in biases]
        n = len(kernel_sizes)
        kz = (C.c_int32 * n)(*[int(k) for k in kernel_sizes])
        ch = (C.c_int32 * n)(*[int(c) for c in channels])
        out = torch.empty(bz, sum(channels), L, dtype=F32, device=t.device)
        check(_lib.lib().rbr_conv_shift_add_fwd(bz, L, n, kz, ch, dev_ptr(t, F32, "T"), ptr_array(bs, F32, "bias"), dev_ptr(out, F32, "out"),
                                                current_stream()), "rbr_conv_shift_add_fwd")
        ctx.job = (bz, L, tuple(kernel_sizes), tuple(channels), tuple(t.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        bz, L, kernel_sizes, channels, tshape = ctx.job
        n = len(kernel_sizes)
        kz = (C.c_int32 * n)(*kernel_sizes)
        ch = (C.c_int32 * n)(*channels)
        d_out = d_out.contiguous()
        dT = torch.empty(tshape, dtype=F32, device=d_out.device)
        dbs = [torch.empty(c, dtype=F32, device=d_out.device) for c in channels]
        check(_lib.lib().rbr_conv_shift_add_bwd(bz, L, n, kz, ch, dev_ptr(d_out, F32, "d_out"), dev_ptr(dT, F32, "dT"),
                                                ptr_array(dbs, F32, "dbias"), current_stream()), "rbr_conv_shift_add_bwd")
        return (dT, None, None, None, None, *dbs)


def conv_shift_add(t, bz, L, kernel_sizes, channels, biases):
    """MyConv1d.forward's epilogue (deepconn/layers.py:46-60): 'same' padding, shifted adds, bias, channel cat, N x C x L layout --
    one launch.  t [bz * L, sum kz*ch] is the layer's product (functional.linear); returns [bz, sum ch, L]."""
    return _ConvShiftAdd.apply(t, bz, L, kernel_sizes, channels, *biases)


# --------------------------------------------------------------------------- SimpleSiamese encoder
class _ReviewBag(torch.autograd.Function):
    """out[n_rev, D] = drop * masked mean of table[ids]  -- rbr_review_bag_* in rbr_hip.h."""

    @staticmethod
    def forward(ctx, table, ids, mask, drop, padding_idx):
        dev_ptr(table.contiguous(), F32, "word table")
        if ids.dim() != 2:
            raise RuntimeError("ids must be [n_reviews, review_len]")
        n_rev, T = ids.shape
        D = table.shape[1]
        L_ = _lib.lib()
        table_c, ids = table.contiguous(), ids.contiguous()
        mask8 = _mask_u8(mask)
        if mask8 is not None and mask8.shape != ids.shape:
            raise AssertionError("input_masks must be [n_reviews, review_len]")
        drop = drop.contiguous() if drop is not None else None
        out = torch.empty(n_rev, D, dtype=F32, device=table.device)
        inv_len = torch.empty(n_rev, dtype=F32, device=table.device)
        check(L_.rbr_review_bag_fwd(n_rev, T, D, dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                    dev_ptr(table_c, F32, "word table"), dev_ptr(drop, F32, "drop"), dev_ptr(out, F32, "out"),
                                    dev_ptr(inv_len, F32, "inv_len"), current_stream()), "rbr_review_bag_fwd")
        ctx.dims = (n_rev, T, D, -1 if padding_idx is None else int(padding_idx), tuple(table.shape))
        ctx.has_mask, ctx.has_drop = mask8 is not None, drop is not None
        ctx.save_for_backward(ids, inv_len, *([mask8] if mask8 is not None else []), *([drop] if drop is not None else []))
        return out

    @staticmethod
    def backward(ctx, d_out):
        n_rev, T, D, pad, shape = ctx.dims
        saved = list(ctx.saved_tensors)
        ids, inv_len = saved[:2]
        k = 2
        mask8 = drop = None
        if ctx.has_mask:
            mask8 = saved[k]; k += 1
        if ctx.has_drop:
            drop = saved[k]
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        dtable = torch.zeros(shape, dtype=F32, device=d_out.device)
        d_out = d_out.contiguous()
        L_ = _lib.lib()
        ws = torch.empty(L_.rbr_review_bag_bwd_ws_bytes(n_rev, T), dtype=torch.uint8, device=d_out.device)
        check(L_.rbr_review_bag_bwd(n_rev, T, D, shape[0], dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                    dev_ptr(drop, F32, "drop"), dev_ptr(inv_len, F32, "inv_len"),
                                    dev_ptr(d_out, F32, "d_out"), pad, dev_ptr(dtable, F32, "dtable"), ws.data_ptr(),
                                    current_stream()), "rbr_review_bag_bwd")
        return dtable, None, None, None, None


def review_bag(table, ids, mask, *, drop=None, padding_idx=0):
    """WordEmbedding -> VariationalDropout (multiplier `drop` [n_rev, D]) -> MaskedAvgPooling1d.  [n_rev, D]."""
    return _ReviewBag.apply(table, ids, mask, drop, padding_idx)


class _AdditiveAttn(torch.autograd.Function):
    """out[B,H], scores[B,R] = AddictiveAttention(node_drop * rev[B,R,H], mask[B,R])  -- rbr_additive_attn_*."""

    @staticmethod
    def forward(ctx, rev, mask, node_drop, Wp, bp, wi):
        if rev.dim() != 3:
            raise RuntimeError("inputs must be [bz, seq_len, hdim]")
        B, R, H = rev.shape
        K = Wp.shape[0]
        L_ = _lib.lib()
        rev, Wp, bp, wi = rev.contiguous(), Wp.contiguous(), bp.contiguous(), wi.contiguous().view(-1)
        mask8 = _mask_u8(mask)
        if mask8 is not None and mask8.dim() != 2:
            raise AssertionError("input_masks.dim() == 2")          # simple_siamese/layers.py:189
        node_drop = node_drop.contiguous() if node_drop is not None else None
        dev = rev.device
        out = torch.empty(B, H, dtype=F32, device=dev)
        scores = torch.empty(B, R, dtype=F32, device=dev)
        t = torch.empty(B, R, K, dtype=F32, device=dev)
        check(L_.rbr_additive_attn_fwd(B, R, H, K, dev_ptr(rev, F32, "inputs"), dev_ptr(mask8, U8, "mask"),
                                       dev_ptr(node_drop, F32, "node_drop"), dev_ptr(Wp, F32, "proj weight"),
                                       dev_ptr(bp, F32, "proj bias"), dev_ptr(wi, F32, "inner_product weight"),
                                       dev_ptr(out, F32, "out"), dev_ptr(scores, F32, "scores"), dev_ptr(t, F32, "t"),
                                       current_stream()), "rbr_additive_attn_fwd")
        ctx.dims = (B, R, H, K)
        ctx.has_mask, ctx.has_nd = mask8 is not None, node_drop is not None
        ctx.wi_shape = None
        ctx.save_for_backward(rev, Wp, wi, scores, t, *([mask8] if mask8 is not None else []),
                              *([node_drop] if node_drop is not None else []))
        ctx.mark_non_differentiable(scores)
        return out, scores

    @staticmethod
    def backward(ctx, d_out, _d_scores):
        B, R, H, K = ctx.dims
        saved = list(ctx.saved_tensors)
        rev, Wp, wi, scores, t = saved[:5]
        k = 5
        mask8 = node_drop = None
        if ctx.has_mask:
            mask8 = saved[k]; k += 1
        if ctx.has_nd:
            node_drop = saved[k]
        L_ = _lib.lib()
        dev = rev.device
        d_out = d_out.contiguous()
        d_rev = torch.empty_like(rev)
        d_Wp = torch.empty_like(Wp)
        d_bp = torch.empty(K, dtype=F32, device=dev)
        d_wi = torch.empty(K, dtype=F32, device=dev)
        ws = torch.empty(L_.rbr_additive_attn_bwd_ws_floats(B, R, H, K), dtype=F32, device=dev)
        check(L_.rbr_additive_attn_bwd(B, R, H, K, dev_ptr(rev, F32, "inputs"), dev_ptr(mask8, U8, "mask"),
                                       dev_ptr(node_drop, F32, "node_drop"), dev_ptr(Wp, F32, "proj weight"),
                                       dev_ptr(wi, F32, "inner_product weight"), dev_ptr(scores, F32, "scores"),
                                       dev_ptr(t, F32, "t"), dev_ptr(d_out, F32, "d_out"), dev_ptr(d_rev, F32, "d_inputs"),
                                       dev_ptr(d_Wp, F32, "d_Wp"), dev_ptr(d_bp, F32, "d_bp"), dev_ptr(d_wi, F32, "d_wi"),
                                       dev_ptr(ws, F32, "ws"), current_stream()), "rbr_additive_attn_bwd")
        return d_rev, None, None, d_Wp, d_bp, d_wi.view(1, K)


def additive_attention(rev, mask, Wp, bp, wi, *, node_drop=None):
    """AddictiveAttention over the reviews (+ NodeDropout multiplier [B,R]).  Returns (out [B,H], scores [B,R,1])."""
    out, scores = _AdditiveAttn.apply(rev, mask, node_drop, Wp, bp, wi)
    return out, scores.unsqueeze(2)


# --------------------------------------------------------------------------- standalone embedding
class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, padding_idx):
        L_ = _lib.lib()
        table = table.contiguous()
        flat = ids.contiguous().view(-1)
        D = table.shape[1]
        out = torch.empty(flat.numel(), D, dtype=F32, device=table.device)
        if flat.numel():
            check(L_.rbr_embedding_fwd(flat.numel(), D, dev_ptr(flat, I64, "ids"), dev_ptr(table, F32, "table"),
                                       dev_ptr(out, F32, "out"), current_stream()), "rbr_embedding_fwd")
        ctx.pad = -1 if padding_idx is None else int(padding_idx)
        ctx.shape = table.shape
        ctx.save_for_backward(flat)
        return out.view(*ids.shape, D)

    @staticmethod
    def backward(ctx, d_out):
        (flat,) = ctx.saved_tensors
        L_ = _lib.lib()
        D = ctx.shape[1]
        dtable = torch.zeros(ctx.shape, dtype=F32, device=d_out.device)
        d_out = d_out.contiguous().view(-1, D)
        if flat.numel():
            check(L_.rbr_embedding_bwd(flat.numel(), D, dev_ptr(flat, I64, "ids"), dev_ptr(d_out, F32, "d_out"), ctx.pad,
                                       dev_ptr(dtable, F32, "dtable"), current_stream()), "rbr_embedding_bwd")
        return dtable, None, None


def embedding(table, ids, padding_idx=0):
    """Materialised row gather table[ids] -> [*ids.shape, D] (WordEmbedding.forward)."""
    return _Embedding.apply(table, ids, padding_idx)


# --------------------------------------------------------------------------- HierPooling
class _HierPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, mask, k, relu, padding_idx):
        n_docs, L = ids.shape
        V, D = table.shape
        L_ = _lib.lib()
        table = table.contiguous()
        ids = ids.contiguous()
        mask8 = _mask_u8(mask)
        pooled = torch.empty(n_docs, D, dtype=F32, device=table.device)
        argmax = torch.empty(n_docs, D, dtype=I32, device=table.device)
        check(L_.rbr_hier_pool_fwd(n_docs, L, D, int(k), dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                   dev_ptr(table, F32, "table"), int(relu), dev_ptr(pooled, F32, "pooled"),
                                   dev_ptr(argmax, I32, "argmax"), current_stream()), "rbr_hier_pool_fwd")
        ctx.args = (n_docs, L, D, int(k), int(relu), -1 if padding_idx is None else int(padding_idx), tuple(table.shape))
        ctx.has_mask = mask8 is not None
        ctx.save_for_backward(ids, argmax, pooled, *([mask8] if mask8 is not None else []))
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        n_docs, L, D, k, relu, pad, shape = ctx.args
        ids, argmax, pooled = ctx.saved_tensors[:3]
        mask8 = ctx.saved_tensors[3] if ctx.has_mask else None
        L_ = _lib.lib()
        dtable = torch.zeros(shape, dtype=F32, device=d_pooled.device)
        d_pooled = d_pooled.contiguous()
        check(L_.rbr_hier_pool_bwd(n_docs, L, D, k, dev_ptr(ids, I64, "ids"), dev_ptr(mask8, U8, "mask"),
                                   dev_ptr(argmax, I32, "argmax"), dev_ptr(pooled, F32, "pooled"),
                                   dev_ptr(d_pooled, F32, "d_pooled"), relu, pad, dev_ptr(dtable, F32, "dtable"),
                                   current_stream()), "rbr_hier_pool_bwd")
        return dtable, None, None, None, None, None


def hier_pool(table, ids, masks, kernel_size, proj_w, proj_b, padding_idx=0):
    """NgramFeat arch="HierPooling": window mean -> global max -> optional Linear -> ReLU.  [n_docs, H]."""
    if proj_w is None:
        return _HierPool.apply(table, ids, masks, kernel_size, True, padding_idx)
    pooled = _HierPool.apply(table, ids, masks, kernel_size, False, padding_idx)
    return linear(pooled, proj_w, proj_b, relu=True)


# --------------------------------------------------------------------------- D-ATT gates
class _DattGate(torch.autograd.Function):
    """gate[B,L] of LocalAttention (win odd) or GlobalAttention (is_global) -- rbr_datt_*_gate_* in rbr_hip.h."""

    @staticmethod
    def forward(ctx, table, w, b0, ids, is_global, padding_idx, rows=None):
        ctx.fanout_acc = _fanout_acc(table)
        ctx.gate_ws = None
        ctx.rows = rows                              # the tower's distinct-token rows (datt_token_rows) or None
        B, L = ids.shape
        E = table.shape[1]
        win = w.shape[2]
        L_ = _lib.lib()
        table, w, b0, ids = table.contiguous(), w.contiguous(), b0.contiguous(), ids.contiguous()
        gate = torch.empty(B, L, dtype=F32, device=table.device)
        if is_global:
            if win != L:
                raise RuntimeError(f"GlobalAttention weight spans {win} positions but documents have {L}")
            check(L_.rbr_datt_global_gate_fwd(B, L, E, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "table"),
                                              dev_ptr(w, F32, "w"), dev_ptr(b0, F32, "b0"), dev_ptr(gate, F32, "gate"),
                                              current_stream()), "rbr_datt_global_gate_fwd")
        else:
            V = table.shape[0]
            ws_bytes = L_.rbr_datt_local_gate_prod_ws_bytes(B, L, E, win, V)     # > 0: token-product form of the gate
            if ws_bytes:
                ctx.gate_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=table.device)
                check(L_.rbr_datt_local_gate_fwd_prod(B, L, E, win, V, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "table"),
                                                      dev_ptr(w, F32, "w"), dev_ptr(b0, F32, "b0"), dev_ptr(gate, F32, "gate"),
                                                      ctx.gate_ws.data_ptr(), None if rows is None else rows.data_ptr(),
                                                      current_stream()), "rbr_datt_local_gate_fwd_prod")
            else:
                check(L_.rbr_datt_local_gate_fwd(B, L, E, win, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "table"),
                                                 dev_ptr(w, F32, "w"), dev_ptr(b0, F32, "b0"), dev_ptr(gate, F32, "gate"),
                                                 current_stream()), "rbr_datt_local_gate_fwd")
        ctx.args = (B, L, E, win, bool(is_global), -1 if padding_idx is None else int(padding_idx))
        ctx.save_for_backward(table, w, ids, gate)
        return gate

    @staticmethod
    def backward(ctx, dgate):
        B, L, E, win, is_global, pad = ctx.args
        table, w, ids, gate = ctx.saved_tensors
        L_ = _lib.lib()
        dev = table.device
        dgate = dgate.contiguous()
        dw = torch.empty_like(w)
        db0 = torch.empty(1, dtype=F32, device=dev)
        acc = _table_acc(ctx.fanout_acc, dev) if ctx.needs_input_grad[0] else None      # the step's shared gradient buffer
        if ctx.gate_ws is not None:      # token-product local gate: the whole table gradient is overwritten (or its rows added)
            dtable = (acc if acc is not None else torch.empty_like(table)) if ctx.needs_input_grad[0] else None
            check(L_.rbr_datt_local_gate_bwd_prod(B, L, E, win, table.shape[0], dev_ptr(ids, I64, "ids"),
                                                  dev_ptr(table, F32, "table"), dev_ptr(w, F32, "w"), dev_ptr(gate, F32, "gate"),
                                                  dev_ptr(dgate, F32, "dgate"), pad, dev_ptr(dw, F32, "dw"),
                                                  dev_ptr(db0, F32, "db0"), dev_ptr(dtable, F32, "dtable"),
                                                  ctx.gate_ws.data_ptr(), None if ctx.rows is None else ctx.rows.data_ptr(),
                                                  int(acc is not None), current_stream()), "rbr_datt_local_gate_bwd_prod")
            return dtable, dw, db0, None, None, None, None
        V = table.shape[0]
        rows_floats = L_.rbr_datt_global_gate_bwd_rows_ws_floats(B, L, E, V) if (ctx.rows is not None and is_global) else 0
        if rows_floats:                  # global gate over the tower's token rows: occurrence matrix, dtable overwritten
            dtable = (acc if acc is not None else torch.empty_like(table)) if ctx.needs_input_grad[0] else None
            ws = torch.empty(rows_floats, dtype=F32, device=dev)
            check(L_.rbr_datt_global_gate_bwd_rows(B, L, E, V, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "table"),
                                                   dev_ptr(w, F32, "w"), dev_ptr(gate, F32, "gate"), dev_ptr(dgate, F32, "dgate"),
                                                   pad, dev_ptr(dw, F32, "dw"), dev_ptr(db0, F32, "db0"),
                                                   dev_ptr(dtable, F32, "dtable"), dev_ptr(ws, F32, "ws"), ctx.rows.data_ptr(),
                                                   int(acc is not None), current_stream()), "rbr_datt_global_gate_bwd_rows")
            return dtable, dw, db0, None, None, None, None
        dtable = torch.zeros_like(table) if ctx.needs_input_grad[0] else None
        ws = torch.empty(max(1, L_.rbr_datt_gate_bwd_ws_floats(B, L, E, win, int(is_global))), dtype=F32, device=dev)
        if is_global:
            check(L_.rbr_datt_global_gate_bwd(B, L, E, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "table"),
                                              dev_ptr(w, F32, "w"), dev_ptr(gate, F32, "gate"), dev_ptr(dgate, F32, "dgate"),
                                              pad, dev_ptr(dw, F32, "dw"), dev_ptr(db0, F32, "db0"),
                                              dev_ptr(dtable, F32, "dtable"), dev_ptr(ws, F32, "ws"), current_stream()),
                  "rbr_datt_global_gate_bwd")
        else:
            check(L_.rbr_datt_local_gate_bwd(B, L, E, win, dev_ptr(ids, I64, "ids"), dev_ptr(table, F32, "table"),
                                             dev_ptr(w, F32, "w"), dev_ptr(gate, F32, "gate"), dev_ptr(dgate, F32, "dgate"),
                                             pad, dev_ptr(dw, F32, "dw"), dev_ptr(db0, F32, "db0"),
                                             dev_ptr(dtable, F32, "dtable"), dev_ptr(ws, F32, "ws"), current_stream()),
                  "rbr_datt_local_gate_bwd")
        return dtable, dw, db0, None, None, None, None


def datt_gate(table, w, b0, ids, *, is_global, padding_idx=0, rows=None):
    """sigmoid attention gate [B,L]; w is the Conv1d weight [1,E,win] (local) or [1,E,L] (global).  `rows`: the tower's
    distinct-token rows (datt_token_rows), shared by its two gates -- the token-product local gate uses them instead of
    building its own, the global gate's backward builds the table gradient from the occurrence matrix over them."""
    return _DattGate.apply(table, w, b0, ids, is_global, padding_idx, rows)


# --------------------------------------------------------------------------- D-ATT: both towers through every kernel in one pass
@contextlib.contextmanager
def _pair_region():
    """with _pair_region() as region: problem 0's calls; region.next(); problem 1's calls -- the C-ABI calls inside are recorded,
    and leave at the end as shared launches (rbr_pair_begin / _next / _end of rbr_hip.h).  Nothing but library calls and
    allocations may happen inside (a torch op would run ahead of the recorded launches)."""
    L_ = _lib.lib()
    check(L_.rbr_pair_begin(), "rbr_pair_begin")

    class _Region:
        paired = singles = 0

        @staticmethod
        def next():
            check(L_.rbr_pair_next(), "rbr_pair_next")

    region = _Region()
    try:
        yield region
    except BaseException:
        L_.rbr_pair_abort()
        raise
    n_p, n_s = C.c_int32(0), C.c_int32(0)
    check(L_.rbr_pair_end(C.byref(n_p), C.byref(n_s)), "rbr_pair_end")
    region.paired, region.singles = n_p.value, n_s.value


PAIR_STATS = {"paired": 0, "singles": 0}      # launches of the last paired D-ATT forward + backward that left as pairs / singly


class _NoRegion:
    """Stands in for a pair region in bench.py's kernel-timing pass (TIMER.enabled): the same calls, launched where they are made,
    so that HIP events can bracket them -- the long kernels of the step (GEMM, gather, the G chain) leave singly in a region too
    (PairSolo), so what the events see is what a region launches."""
    paired = singles = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    @staticmethod
    def next():
        pass


def _region_or_timed():
    return _NoRegion() if TIMER.enabled else _pair_region()


def _timed_call(name, rc_fn):
    """check(rc_fn()) bracketed by TIMER events when the timing pass is on."""
    ev = TIMER.record(name)
    check(rc_fn(), name)
    if ev is not None:
        ev.record()


def datt_pair_applies(table, docs2, local_w, conv_ws) -> bool:
    """True when both D-ATT towers can go through datt_towers(): equal shapes by construction (one [2B, L] id block), the
    token-product forms of the local gate and of the merged four-bank conv apply, the distinct-token rows pay, and the GEMM is
    one of the bf16-plane kernels (the f32 MFMA kernel still launches directly)."""
    if not (table.is_cuda and table.dtype == F32 and docs2.dim() == 2 and docs2.shape[0] % 2 == 0):
        return False
    if os.environ.get("RBR_DATT_PAIRED", "1") == "0" or get_prod_precision() == "f32":
        return False
    B, L = docs2.shape[0] // 2, docs2.shape[1]
    V, E = table.shape
    win = int(local_w.shape[2])
    L_ = _lib.lib()
    if V * 5 > B * L * 2 or B * L < 4096 or os.environ.get("RBR_DATT_ROWS", "1") == "0":
        return False
    if not L_.rbr_datt_local_gate_prod_ws_bytes(B, L, E, win, V) or not L_.rbr_datt_global_gate_bwd_rows_ws_floats(B, L, E, V):
        return False
    desc = _lib.make_desc(B, L, E, V, [int(w.shape[2]) for w in conv_ws], [int(w.shape[0]) for w in conv_ws], PAD_VALID, ACT_TANH, 0,
                          _lib.conv_gate_split(1))
    return bool(L_.rbr_textcnn_fwd_ws_bytes(C.byref(desc)) > 0 and L_.rbr_textcnn_bwd_prod_ws_bytes(C.byref(desc)) > 0
                and conv_ws[0].shape[2] == 1)


class _DattTowers(torch.autograd.Function):
    """feats [2B, l_out + 3 g_out] = both D-ATT towers' encoders (reference models/dual_att/dual_att.py:45-57, layers.py:43-53,
    81-89): per tower the distinct-token rows, the local gate, the global gate and the merged four-bank gated conv (local 1-wide
    bank under the local gate, global 2/3/4-wide banks under the global gate) -- the SAME C-ABI calls the single-tower functions
    make (_DattGate, _TextCNN), issued for tower 0 and tower 1 inside a pair region, so every kernel runs once for both towers
    (gridDim.z = 2).  The towers have separate parameters and share the word table; each tower adds its table-gradient rows
    into a buffer of its own (two towers' workgroups of one launch may meet in a row) and the buffers are summed at the end."""

    N_TOWER_PARAMS = 12        # local attn w, b | global attn w, b | conv weights x4 | conv biases x4

    @staticmethod
    def forward(ctx, table, docs2, padding_idx, pad_runs, *params):
        NP = _DattTowers.N_TOWER_PARAMS
        if len(params) != 2 * NP:
            raise RuntimeError("datt_towers: expected 2 x 12 tower parameters")
        L_ = _lib.lib()
        dev = table.device
        table_c = table.contiguous()
        dev_ptr(table_c, F32, "word table")
        docs2 = docs2.contiguous()
        dev_ptr(docs2, I64, "docs")
        B, L = docs2.shape[0] // 2, docs2.shape[1]
        V, E = table.shape
        pad = -1 if padding_idx is None else int(padding_idx)
        st = current_stream()
        towers = []
        for t in range(2):
            q = [x.contiguous() for x in params[t * NP:(t + 1) * NP]]
            tw = _ConvSaved(ids=docs2[t * B:(t + 1) * B], lw=q[0], lb=q[1], gw=q[2], gb=q[3], ws=q[4:8], bs=q[8:12])
            if tw.gw.shape[2] != L:
                raise RuntimeError(f"GlobalAttention weight spans {tw.gw.shape[2]} positions but documents have {L}")
            towers.append(tw)
        win = int(towers[0].lw.shape[2])
        kz = [int(w.shape[2]) for w in towers[0].ws]
        ch = [int(w.shape[0]) for w in towers[0].ws]
        for tw in towers:
            if [int(w.shape[2]) for w in tw.ws] != kz or [int(w.shape[0]) for w in tw.ws] != ch or int(tw.lw.shape[2]) != win:
                raise RuntimeError("datt_towers: the towers' layers must have equal shapes")
        flags = _lib.conv_gate_split(1) | (_lib.CONV_PAD_RUNS if (pad_runs and padding_idx is not None
                                                                   and os.environ.get("RBR_PAD_RUNS", "1") != "0") else 0)
        desc = _lib.make_desc(B, L, E, V, kz, ch, PAD_VALID, ACT_TANH, padding_idx, flags)
        Ctot = sum(ch)
        n_part = L_.rbr_textcnn_partial_elems(C.byref(desc))
        ws_bytes = L_.rbr_textcnn_fwd_ws_bytes(C.byref(desc))
        rows_bytes = L_.rbr_datt_token_rows_ws_bytes(B, L, V)
        gws_bytes = L_.rbr_datt_local_gate_prod_ws_bytes(B, L, E, win, V)
        if not (n_part and ws_bytes and rows_bytes and gws_bytes):
            check(-1, "datt_towers plan (datt_pair_applies() was not consulted)")
        feats = torch.empty(2 * B, Ctot, dtype=F32, device=dev)
        argmax = torch.empty(2 * B, Ctot, dtype=I32, device=dev)
        for t, tw in enumerate(towers):          # allocations only: nothing here launches
            tw.rows = torch.empty(rows_bytes, dtype=torch.uint8, device=dev)
            tw.gate2 = torch.empty(2, B, L, dtype=F32, device=dev)          # plane 0: local gate, plane 1: global gate
            tw.gate_ws = torch.empty(gws_bytes, dtype=torch.uint8, device=dev)
            tw.pval = torch.empty(n_part, dtype=F32, device=dev)
            tw.pidx = torch.empty(n_part, dtype=I32, device=dev)
            tw.prod_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            tw.feat, tw.argmax = feats[t * B:(t + 1) * B], argmax[t * B:(t + 1) * B]
        ev = TIMER.record("datt_towers_fwd")
        # (Measured and dropped, round 4: tower 1's product-table GEMM on a second stream beside tower 0's gather -- MFMA / LDS work
        # beside L2-request work.  The GEMM's 137 MB of product-table writes push the table the gather is reading out of the L2s and
        # the Infinity Cache: the gather 125 -> 205 us, the GEMM 48 -> 150 us, the step +80 us.  Also dropped: both gates on the second
        # stream beside tower 0's GEMM, which needs the token list only -- the GEMM 49 -> 81 us, the gates' three kernels 102 -> 134,
        # the gather starts 4 us earlier: everything here queues at the same L2s.)
        with _region_or_timed() as region:
            for t, tw in enumerate(towers):
                if t == 1:
                    region.next()
                ids_p = dev_ptr(tw.ids, I64, "ids")
                check(L_.rbr_datt_token_rows(B, L, V, ids_p, tw.rows.data_ptr(), st), "rbr_datt_token_rows")
                check(L_.rbr_datt_local_gate_fwd_prod(B, L, E, win, V, ids_p, dev_ptr(table_c, F32, "table"), dev_ptr(tw.lw, F32, "w"),
                                                      dev_ptr(tw.lb, F32, "b0"), dev_ptr(tw.gate2[0], F32, "gate"), tw.gate_ws.data_ptr(),
                                                      tw.rows.data_ptr(), st), "rbr_datt_local_gate_fwd_prod")
                check(L_.rbr_datt_global_gate_fwd(B, L, E, ids_p, dev_ptr(table_c, F32, "table"), dev_ptr(tw.gw, F32, "w"),
                                                  dev_ptr(tw.gb, F32, "b0"), dev_ptr(tw.gate2[1], F32, "gate"), st),
                      "rbr_datt_global_gate_fwd")
                wsp = tw.prod_ws.data_ptr()
                check(L_.rbr_textcnn_prod_prepare(C.byref(desc), ids_p, None, ptr_array(tw.ws, F32, "conv weight"),
                                                  dev_ptr(tw.pidx, I32, "pidx"), wsp, st), "rbr_textcnn_prod_prepare")
                _timed_call("textcnn_prod_table", lambda: L_.rbr_textcnn_prod_table(C.byref(desc), dev_ptr(table_c, F32, "word table"), wsp, st))
                _timed_call("textcnn_prod_pool", lambda: L_.rbr_textcnn_prod_pool(
                    C.byref(desc), ids_p, None, dev_ptr(tw.gate2, F32, "gate"), dev_ptr(tw.pval, F32, "pval"),
                    dev_ptr(tw.pidx, I32, "pidx"), wsp, st))
                check(L_.rbr_textcnn_pool_finalize(C.byref(desc), dev_ptr(tw.pval, F32, "pval"), dev_ptr(tw.pidx, I32, "pidx"),
                                                   ptr_array(tw.bs, F32, "conv bias"), dev_ptr(tw.feat, F32, "feat"),
                                                   dev_ptr(tw.argmax, I32, "argmax"), st), "rbr_textcnn_pool_finalize")
        if ev is not None:
            ev.record()
        PAIR_STATS["paired"], PAIR_STATS["singles"] = region.paired, region.singles
        for tw in towers:
            tw.pval = tw.pidx = None           # only the backward's inputs stay
        ctx.towers, ctx.desc, ctx.table = towers, desc, table_c
        ctx.dims = (B, L, V, E, win, pad)
        ctx.keep = (feats, argmax, docs2)
        ctx.set_materialize_grads(False)
        return feats

    @staticmethod
    def backward(ctx, d_feats):
        NP = _DattTowers.N_TOWER_PARAMS
        towers, desc, table = ctx.towers, ctx.desc, ctx.table
        B, L, V, E, win, pad = ctx.dims
        L_ = _lib.lib()
        dev = table.device
        if d_feats is None:
            d_feats = torch.zeros_like(ctx.keep[0])
        d_feats = d_feats.contiguous()
        need_table = ctx.needs_input_grad[0]
        # Table-gradient buffers [4, V, E], allocated (and the gates' cleared) BEFORE the regions: [t] receives tower t's gate rows
        # (added: local gate, then global gate, one after the other on this stream), [2 + t] its conv rows (g_times_w's dense
        # form: every row written, absent tokens' rows zero).  Gates and conv of a tower run on DIFFERENT streams below, and two
        # towers' workgroups of one launch may meet in a row: four writers, four buffers, summed once at the end.
        dtab = torch.empty(4, V, E, dtype=F32, device=dev) if need_table else None
        if need_table:
            dtab[:2].zero_()
        bws_bytes = L_.rbr_textcnn_bwd_prod_ws_bytes(C.byref(desc))
        gg_floats = L_.rbr_datt_global_gate_bwd_rows_ws_floats(B, L, E, V)
        wsn = L_.rbr_textcnn_bwd_ws_floats(C.byref(desc))
        for tw in towers:
            tw.bws = torch.empty(bws_bytes, dtype=torch.uint8, device=dev)
            tw.dgate2 = torch.empty(2, B, L, dtype=F32, device=dev)          # zeroed by the launch that zeroes G's rows
            tw.gg_ws = torch.empty(gg_floats, dtype=F32, device=dev)
            tw.wsb = torch.empty(max(wsn, 1), dtype=F32, device=dev)
            tw.grads = ([torch.empty_like(tw.lw), torch.empty(1, dtype=F32, device=dev), torch.empty_like(tw.gw),
                         torch.empty(1, dtype=F32, device=dev)] + [torch.empty_like(w) for w in tw.ws]
                        + [torch.empty(w.shape[0], dtype=F32, device=dev) for w in tw.ws])
        st = current_stream()
        # Two streams, two regions.  Region A: G and d(gate) of both towers on this stream (zero_g_rows, build_g), the conv weight
        # gradient (dw_partial4, dw_reduce: it starts from d_feats alone) on the second one.  Region B: the gates' backwards on
        # this stream -- they need d(gate) -- and the conv's sparse product G @ Wprod^T on the second stream behind the weight
        # gradient -- it needs G.  (One stream for everything but the weight gradient, as in round 3: the second stream idle for
        # 600 us of the backward's 750.)
        side = _side_stream(dev)
        fork = join = None
        if side is not None:
            fork = torch.cuda.Event()
            fork.record()
            side.wait_event(fork)
        st2 = side.cuda_stream if side is not None else st
        st_prod = st if TIMER.enabled else st2      # (the timing pass brackets calls with events on THIS stream)
        ev = TIMER.record("datt_towers_bwd")
        tab = dev_ptr(table, F32, "table")
        paired = singles = 0
        with _region_or_timed() as region:
            for t, tw in enumerate(towers):
                if t == 1:
                    region.next()
                ids_p = dev_ptr(tw.ids, I64, "ids")
                d_feat = d_feats[t * B:(t + 1) * B]
                check(L_.rbr_textcnn_bwd_dw(C.byref(desc), ids_p, None, dev_ptr(tw.gate2, F32, "gate"), tab,
                                            dev_ptr(tw.feat, F32, "feat"), dev_ptr(tw.argmax, I32, "argmax"),
                                            dev_ptr(d_feat, F32, "d_feat"), ptr_array(tw.grads[4:8], F32, "dW"),
                                            ptr_array(tw.grads[8:12], F32, "dbias"), dev_ptr(tw.wsb, F32, "ws"), st2), "rbr_textcnn_bwd_dw")
                check(L_.rbr_textcnn_bwd_dtable_prod_ex(C.byref(desc), ids_p, None, dev_ptr(tw.gate2, F32, "gate"),
                                                        dev_ptr(tw.feat, F32, "feat"), dev_ptr(tw.argmax, I32, "argmax"),
                                                        dev_ptr(d_feat, F32, "d_feat"), tw.prod_ws.data_ptr(), tw.bws.data_ptr(), None,
                                                        dev_ptr(tw.dgate2, F32, "dgate"), None, _lib.G_BUILD, st), "rbr_textcnn_bwd_g_build")
        paired, singles = paired + region.paired, singles + region.singles
        if side is not None:          # G is complete on this stream (the region's launches have left): the product may follow it
            g_done = torch.cuda.Event()
            g_done.record()
            side.wait_event(g_done)
        with _region_or_timed() as region:
            for t, tw in enumerate(towers):
                if t == 1:
                    region.next()
                ids_p = dev_ptr(tw.ids, I64, "ids")
                dt = dev_ptr(dtab[t], F32, "dtable") if need_table else None
                if need_table:
                    dc = dev_ptr(dtab[2 + t], F32, "dtable")
                    _timed_call("textcnn_bwd_g_product", lambda: L_.rbr_textcnn_bwd_dtable_prod_ex(
                        C.byref(desc), None, None, None, None, None, None, tw.prod_ws.data_ptr(), tw.bws.data_ptr(), dc, None, None,
                        _lib.G_PRODUCT, st_prod))
                check(L_.rbr_datt_local_gate_bwd_prod(B, L, E, win, V, ids_p, tab, dev_ptr(tw.lw, F32, "w"),
                                                      dev_ptr(tw.gate2[0], F32, "gate"), dev_ptr(tw.dgate2[0], F32, "dgate"), pad,
                                                      dev_ptr(tw.grads[0], F32, "dw"), dev_ptr(tw.grads[1], F32, "db0"), dt,
                                                      tw.gate_ws.data_ptr(), tw.rows.data_ptr(), 1, st), "rbr_datt_local_gate_bwd_prod")
                # the global gate's weight phase (dpre, dw: both towers per launch); its table rows follow outside the region
                check(L_.rbr_datt_global_gate_bwd_rows(B, L, E, V, ids_p, tab, dev_ptr(tw.gw, F32, "w"),
                                                       dev_ptr(tw.gate2[1], F32, "gate"), dev_ptr(tw.dgate2[1], F32, "dgate"), pad,
                                                       dev_ptr(tw.grads[2], F32, "dw"), dev_ptr(tw.grads[3], F32, "db0"), None,
                                                       dev_ptr(tw.gg_ws, F32, "ws"), tw.rows.data_ptr(), 1, st),
                      "rbr_datt_global_gate_bwd_rows")
        paired, singles = paired + region.paired, singles + region.singles
        if need_table:
            # ... a chain of four launches over the tower's 120 MB occurrence matrix, one tower at a time either way: tower 0's on
            # this stream into its gate buffer, tower 1's on the second stream behind the sparse products, into the buffer its
            # conv rows were written to there (same stream: no fifth buffer) -- both streams then end about together
            if side is not None and not TIMER.enabled:
                dpre_done = torch.cuda.Event()
                dpre_done.record()
                side.wait_event(dpre_done)
            for t, tw in enumerate(towers):
                on_side = t == 1 and side is not None and not TIMER.enabled
                check(L_.rbr_datt_global_gate_bwd_rows(B, L, E, V, dev_ptr(tw.ids, I64, "ids"), None, dev_ptr(tw.gw, F32, "w"), None, None,
                                                       pad, None, None, dev_ptr(dtab[3] if on_side else dtab[t], F32, "dtable"),
                                                       dev_ptr(tw.gg_ws, F32, "ws"), tw.rows.data_ptr(), 1 | 2, st2 if on_side else st),
                      "rbr_datt_global_gate_bwd_rows")
        if side is not None:
            with torch.cuda.stream(side):
                join = torch.cuda.Event()
                join.record()
            _join(join)
        if ev is not None:
            ev.record()
        PAIR_STATS["paired"] += paired
        PAIR_STATS["singles"] += singles
        dtable = dtab.sum(0) if need_table else None
        grads = towers[0].grads + towers[1].grads
        for tw in towers:
            tw.bws = tw.dgate2 = tw.gg_ws = tw.wsb = tw.grads = None
        return (dtable, None, None, None, *grads)


def datt_towers(table, docs2, u_params, i_params, *, padding_idx=0, pad_runs=True):
    """Both D-ATT towers' encoders in one pass (see _DattTowers): docs2 [2B, L] int64 (user documents, then item documents);
    u_params / i_params = (local attn weight [1,E,win], bias [1], global attn weight [1,E,L], bias [1], [4 conv weights],
    [4 conv biases]).  Returns feats [2B, l_out + 3 g_out], user rows first: cat((local, global), 1) per tower."""
    flat = []
    for q in (u_params, i_params):
        lw, lb, gw, gb, ws, bs = q
        flat += [lw, lb, gw, gb, *ws, *bs]
    return _DattTowers.apply(table, docs2, padding_idx, pad_runs, *flat)


def datt_token_rows(ids, vocab_size):
    """Distinct-token row maps of one tower's documents (rbr_datt_token_rows), or None where they do not pay (few positions
    per vocabulary entry: the same rule as the token-product conv)."""
    B, L = ids.shape
    if vocab_size * 5 > B * L * 2 or B * L < 4096 or os.environ.get("RBR_DATT_ROWS", "1") == "0":
        return None
    L_ = _lib.lib()
    rows = torch.empty(L_.rbr_datt_token_rows_ws_bytes(B, L, vocab_size), dtype=torch.uint8, device=ids.device)
    ids = ids.contiguous()
    check(L_.rbr_datt_token_rows(B, L, vocab_size, dev_ptr(ids, I64, "ids"), rows.data_ptr(), current_stream()),
          "rbr_datt_token_rows")
    return rows
