"""Dev: run the cfg2 conv kernel from a -DRBR_DIAG build (built on the fly into tools/diag/) and print the share of
wave-0 cycles each segment of the item loop takes (s_memtime stamps).  Shares only: stamps cost ~40 cycles.

    python tools/dev_conv_diag.py [--nomask] [--product]     # --product: the distinct-token GEMM (store mode)
"""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, synth
from review_based_recommender_amd import _lib

csrc = os.path.join(ROOT, "review-based-recommender_amd", "csrc")
variant = os.environ.get("RBR_DIAG_VARIANT", "")      # 1: no weight DMA in the loop, 2: no LDS weight reads, 3: no barriers (results wrong)
out = os.path.join(ROOT, "tools", "diag", f"librbr_diag{variant}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = sorted(glob.glob(os.path.join(csrc, "*.hip")))
if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in srcs + glob.glob(os.path.join(csrc, "*.h"))):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRBR_DIAG", *([f"-DRBR_DIAG_VARIANT={variant}"] if variant else []),
                           "-o", out] + srcs)
lib = C.CDLL(out)
for n in ("rbr_textcnn_pack", "rbr_textcnn_conv_fwd", "rbr_textcnn_packed_floats", "rbr_textcnn_partial_elems",
          "rbr_textcnn_fwd_ws_bytes", "rbr_textcnn_prod_prepare", "rbr_textcnn_prod_table", "rbr_set_conv_mode"):
    getattr(lib, n).restype, getattr(lib, n).argtypes = _lib.SIGNATURES[n]
lib.rbr_diag_fetch.restype, lib.rbr_diag_fetch.argtypes = C.c_int, [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS["cfg2"]
p = synth.deepconn_params(cfg, 0); b = synth.deepconn_batch(cfg, 1)
ws = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.weight"].to(dev) for i in range(3)]
table = p["word_embeddings.embedding.weight"].to(dev)
ids = torch.cat([b["u_docs"], b["i_docs"]]).to(dev); mask = torch.cat([b["u_masks"], b["i_masks"]]).to(dev).view(torch.uint8)
if "--nomask" in sys.argv: mask = None
product = "--product" in sys.argv
lib.rbr_set_conv_mode(2 if product else 1)
d = _lib.make_desc(ids.shape[0], ids.shape[1], 300, table.shape[0], [3, 5, 7], [50, 50, 50], 0, 0, 0)
npk, npart = lib.rbr_textcnn_packed_floats(C.byref(d)), lib.rbr_textcnn_partial_elems(C.byref(d))
packed = torch.empty(npk, device=dev); pval = torch.zeros(npart, device=dev); pidx = torch.zeros(npart, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
W = _lib.ptr_array(ws, torch.float32, "w")
mp = mask.data_ptr() if mask is not None else None
if product:
    wsb = torch.empty(lib.rbr_textcnn_fwd_ws_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    for _ in range(3):
        assert lib.rbr_textcnn_prod_prepare(C.byref(d), ids.data_ptr(), mp, W, pidx.data_ptr(), wsb.data_ptr(), st) == 0
        assert lib.rbr_textcnn_prod_table(C.byref(d), table.data_ptr(), wsb.data_ptr(), st) == 0
else:
    lib.rbr_textcnn_pack(C.byref(d), W, packed.data_ptr(), st)
    for _ in range(3):
        assert lib.rbr_textcnn_conv_fwd(C.byref(d), ids.data_ptr(), mp, None, table.data_ptr(), None, packed.data_ptr(),
                                        pval.data_ptr(), pidx.data_ptr(), None, st) == 0
torch.cuda.synchronize()
host = np.zeros(8 * 1024, dtype=np.uint64)
assert lib.rbr_diag_fetch(host.ctypes.data, host.size) == 0
diag = host.reshape(1024, 8).astype(np.float64)
diag = diag[diag.sum(1) > 0]
if product and diag[:, 5].max() > 1e6:       # prod_gemm_kernel: slots 5 / 3 hold the absolute start / end stamps
    t0, t1 = diag[:, 5], diag[:, 3]
    print(f"launch span {t1.max() - t0.min():.0f} ticks; start skew {t0.max() - t0.min():.0f}; end skew {t1.max() - t1.min():.0f}; "
          f"median WG lifetime {np.median(t1 - t0):.0f}")
    diag[:, 5] = 0; diag[:, 3] = 0
names = ["item pull", "item prologue", "row gather+barrier", "prefetch issue", "LDS reads + MFMA", "(unused)", "vmcnt + barrier", "epilogue"]
tot = diag.sum(1).mean()
print(f"workgroups {len(diag)}, mean cycles per WG {tot:.0f} (s_memtime ticks at 100 MHz)")
for k, n in enumerate(names):
    print(f"  {n:22s} {100 * diag[:, k].mean() / tot:6.2f} %   ({diag[:, k].mean():.0f})")
