"""ctypes wrapper of oracle/_build/libtextcnn_ref.so (CPU ORACLE -- test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libtextcnn_ref.so")


def load():
    """RBR_C_REF_LIB names another build of the same source (tests/test_asan_host.py: the -fsanitize=address,undefined one)."""
    alt = os.environ.get("RBR_C_REF_LIB")
    if alt:
        return C.CDLL(alt)
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return C.CDLL(LIB)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _pp(arrs):
    arr = (C.POINTER(C.c_float) * len(arrs))()
    for i, a in enumerate(arrs):
        arr[i] = _p(a, C.c_float)
    return arr


def textcnn_fwd(ids, mask, gate, table, Ws, bs, pad_valid=False, act=0):
    lib = load()
    n_docs, L = ids.shape
    D = table.shape[1]
    kz = (C.c_int * len(Ws))(*[w.shape[2] for w in Ws])
    ch = (C.c_int * len(Ws))(*[w.shape[0] for w in Ws])
    Ctot = sum(w.shape[0] for w in Ws)
    feat = np.empty((n_docs, Ctot), np.float32)
    am = np.empty((n_docs, Ctot), np.int32)
    m8 = None if mask is None else np.ascontiguousarray(mask.astype(np.uint8))
    lib.textcnn_fwd_ref(n_docs, L, D, len(Ws), kz, ch, int(pad_valid), int(act), _p(ids, C.c_int64), _p(m8, C.c_uint8),
                        _p(gate, C.c_float), _p(table, C.c_float), _pp(Ws), _pp(bs), _p(feat, C.c_float), _p(am, C.c_int32))
    return feat, am


def textcnn_bwd_sparse(ids, mask, gate, table, Ws, feat, am, d_feat, pad_valid=False, act=0, padding_idx=0):
    lib = load()
    n_docs, L = ids.shape
    V, D = table.shape
    kz = (C.c_int * len(Ws))(*[w.shape[2] for w in Ws])
    ch = (C.c_int * len(Ws))(*[w.shape[0] for w in Ws])
    dWs = [np.empty_like(w) for w in Ws]
    dbs = [np.empty(w.shape[0], np.float32) for w in Ws]
    dtable = np.zeros_like(table)
    dgate = np.zeros((n_docs, L), np.float32) if gate is not None else None
    m8 = None if mask is None else np.ascontiguousarray(mask.astype(np.uint8))
    lib.textcnn_bwd_sparse_ref(n_docs, L, D, V, len(Ws), kz, ch, int(pad_valid), int(act), int(padding_idx),
                               _p(ids, C.c_int64), _p(m8, C.c_uint8), _p(gate, C.c_float), _p(table, C.c_float), _pp(Ws),
                               _p(feat, C.c_float), _p(am, C.c_int32), _p(d_feat, C.c_float), _pp(dWs), _pp(dbs),
                               _p(dtable, C.c_float), _p(dgate, C.c_float))
    return dWs, dbs, dtable, dgate
