"""Fuzz of the HOST side of the C ABI under AddressSanitizer + UBSan (CPU box, no GPU, no torch).

Run by tests/test_asan_host.py as
    LD_PRELOAD=<clang asan runtime> python tests/asan_fuzz_host.py <librbr_hip_hostasan.so>
against a host-only build of csrc/*.hip (`make -C oracle asan`: hipcc --offload-host-only -fsanitize=address,undefined).
What runs on the host in the product library is plan / layout / workspace-size arithmetic and descriptor validation; this
driver throws hostile descriptors at every size function and NULL / hostile arguments at the launch entry points, which must
answer with RBR_ERR_BAD_ARG / RBR_ERR_UNSUPPORTED / a size of 0 -- never with a sanitizer report, a crash or a launch."""
import ctypes as C
import random
import sys

RBR_MAX_WIDTHS = 8


class Desc(C.Structure):
    _fields_ = [("n_docs", C.c_int32), ("L", C.c_int32), ("D", C.c_int32), ("V", C.c_int32), ("n_widths", C.c_int32),
                ("kz", C.c_int32 * RBR_MAX_WIDTHS), ("ch", C.c_int32 * RBR_MAX_WIDTHS),
                ("pad_mode", C.c_int32), ("act", C.c_int32), ("padding_idx", C.c_int32), ("flags", C.c_int32)]


DESC_SIZE_FNS = ["rbr_textcnn_packed_floats", "rbr_textcnn_partial_elems", "rbr_textcnn_fwd_ws_bytes", "rbr_textcnn_bwd_ws_floats",
                 "rbr_textcnn_bwd_prod_ws_bytes", "rbr_textcnn_bwd_dtable_list_ws_bytes", "rbr_textcnn_row_grad_partials",
                 "rbr_textcnn_bwd_dw_from_g_ws_floats", "rbr_textcnn_taps_count"]
INT_SIZE_FNS = {"rbr_pair_head_bwd_ws_floats": 2, "rbr_review_attn_bwd_ws_floats": 4, "rbr_datt_gate_bwd_ws_floats": 5,
                "rbr_datt_local_gate_prod_ws_bytes": 5, "rbr_datt_token_rows_ws_bytes": 3, "rbr_datt_global_gate_bwd_rows_ws_floats": 4,
                "rbr_linear_bwd_ws_floats": 2, "rbr_dedup_ws_bytes": 2, "rbr_review_bag_bwd_ws_bytes": 2,
                "rbr_additive_attn_bwd_ws_floats": 4}
EDGE = [0, 1, -1, 2, 3, 7, 8, 9, 31, 32, 33, 63, 64, 65, 100, 255, 256, 300, 512, 1024, 4096, 50002, 65535, 65536, 1 << 20,
        (1 << 31) - 1, -(1 << 31), 1 << 30, -7]


def rand_desc(rng, sane):
    d = Desc()
    pick = (lambda lo, hi: rng.randint(lo, hi)) if sane else (lambda lo, hi: rng.choice(EDGE))
    d.n_docs, d.L, d.D, d.V = pick(1, 4096), pick(1, 2048), pick(1, 512), pick(2, 100000)
    d.n_widths = rng.randint(1, 8) if sane else rng.choice([0, 1, 3, 8, 9, -1, 100, (1 << 31) - 1])
    for i in range(RBR_MAX_WIDTHS):
        d.kz[i] = rng.randint(1, 9) if sane else rng.choice(EDGE)
        d.ch[i] = rng.randint(1, 300) if sane else rng.choice(EDGE)
    d.pad_mode, d.act = (rng.randint(0, 1), rng.randint(0, 1)) if sane else (rng.choice(EDGE), rng.choice(EDGE))
    d.padding_idx = rng.choice([-1, 0, 1]) if sane else rng.choice(EDGE)
    d.flags = rng.choice([0, 1, 1 << 8, (3 << 8) | 1]) if sane else rng.choice(EDGE)
    return d


def main(path):
    lib = C.CDLL(path)
    lib.rbr_last_error.restype = C.c_char_p
    rng = random.Random(20261005)
    n_calls = 0
    for name in DESC_SIZE_FNS + ["rbr_textcnn_dtable_from_taps_ws_bytes"]:
        getattr(lib, name).restype = C.c_size_t
    lib.rbr_textcnn_taps_owner_rows.restype = C.c_int32
    sizes_seen = 0
    for it in range(4000):
        d = rand_desc(rng, sane=(it % 3 == 0))
        for name in DESC_SIZE_FNS:
            v = getattr(lib, name)(C.byref(d))
            sizes_seen += int(v > 0)
            n_calls += 1
        for n_sets in (rng.choice(EDGE), 1, 8):
            lib.rbr_textcnn_dtable_from_taps_ws_bytes(C.byref(d), C.c_int32(n_sets))
            lib.rbr_textcnn_taps_owner_rows(C.byref(d), C.c_int32(n_sets))
            n_calls += 2
    # NULL descriptor
    for name in DESC_SIZE_FNS:
        assert getattr(lib, name)(None) == 0, name
    for name, n in INT_SIZE_FNS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_size_t
        for _ in range(600):
            fn(*[C.c_int32(rng.choice(EDGE)) for _ in range(n)])
            n_calls += 1
    # launch entry points: NULL pointers with plausible descriptors, hostile descriptors with NULL pointers -> an error code,
    # never a launch (this build has no device code and this box has no GPU)
    st = C.c_void_p(0)
    bad = 0
    for it in range(600):
        d = rand_desc(rng, sane=(it % 2 == 0))
        rcs = [
            lib.rbr_textcnn_pack(C.byref(d), None, None, st),
            lib.rbr_textcnn_prod_prepare(C.byref(d), None, None, None, None, None, st),
            lib.rbr_textcnn_prod_table(C.byref(d), None, None, st),
            lib.rbr_textcnn_prod_pool(C.byref(d), None, None, None, None, None, None, st),
            lib.rbr_textcnn_pool_finalize(C.byref(d), None, None, None, None, None, st),
            lib.rbr_textcnn_bwd_dtable_prod_ex(C.byref(d), None, None, None, None, None, None, None, None, None, None, None,
                                               C.c_int32(rng.choice([0, 1, 2, 3, 8, 10, 31, -1])), st),
            lib.rbr_textcnn_bwd_taps(C.byref(d), None, None, None, None, None, None, None, st),
            lib.rbr_textcnn_dtable_from_taps(C.byref(d), C.c_int32(rng.choice(EDGE)), None, None, None, None, None, st),
            lib.rbr_textcnn_dtable_from_taps_owner(C.byref(d), C.c_int32(rng.choice(EDGE)), C.c_int32(rng.choice(EDGE)), None, None,
                                                   None, None, None, None, st),
        ]
        n_calls += len(rcs)
        for rc in rcs:
            assert rc != 0, "a launch entry point accepted NULL pointers"
            bad += 1
        assert lib.rbr_last_error() is not None
    print(f"ASAN HOST FUZZ OK: {n_calls} calls, {sizes_seen} non-zero sizes, {bad} refusals", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
