"""Dev: run the cfg2 conv kernel from a -DRBR_DIAG build (tools/diag/librbr_diag.so) and print the
share of wave-0 cycles each segment of the item loop takes (s_memtime stamps).  Shares only: stamps cost ~40 cycles."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, synth
from review_based_recommender_amd import _lib
lib = C.CDLL(os.path.join(ROOT, "tools", "diag", "librbr_diag.so"))
for n in ("rbr_textcnn_pack", "rbr_textcnn_conv_fwd", "rbr_textcnn_packed_floats", "rbr_textcnn_partial_elems"):
    getattr(lib, n).restype, getattr(lib, n).argtypes = _lib.SIGNATURES[n]
dev = torch.device("cuda:0")
cfg = synth.DEEPCONN_CFGS["cfg2"]
p = synth.deepconn_params(cfg, 0); b = synth.deepconn_batch(cfg, 1)
ws = [p[f"ngram.feature_layer.0.list_of_conv1d.{i}.weight"].to(dev) for i in range(3)]
table = p["word_embeddings.embedding.weight"].to(dev)
ids = torch.cat([b["u_docs"], b["i_docs"]]).to(dev); mask = torch.cat([b["u_masks"], b["i_masks"]]).to(dev).view(torch.uint8)
if "--nomask" in sys.argv: mask = None
d = _lib.make_desc(ids.shape[0], ids.shape[1], 300, table.shape[0], [3, 5, 7], [50, 50, 50], 0, 0, 0)
npk, npart = lib.rbr_textcnn_packed_floats(C.byref(d)), lib.rbr_textcnn_partial_elems(C.byref(d))
packed = torch.empty(npk, device=dev); pval = torch.zeros(npart + 8 * 2048 * 2, device=dev); pidx = torch.zeros(npart, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.rbr_textcnn_pack(C.byref(d), _lib.ptr_array(ws, torch.float32, "w"), packed.data_ptr(), st)
for _ in range(3):
    rc = lib.rbr_textcnn_conv_fwd(C.byref(d), ids.data_ptr(), mask.data_ptr() if mask is not None else None, None, table.data_ptr(), None, packed.data_ptr(), pval.data_ptr(), pidx.data_ptr(), None, st)
    assert rc == 0
torch.cuda.synchronize()
total_wt = ids.shape[0] * 16
tail = pval[total_wt * 160:].cpu().numpy().view(np.uint64)
nblk = 512
diag = tail[: 8 * nblk].reshape(nblk, 8).astype(np.float64)
diag = diag[diag.sum(1) > 0]
names = ["item pull", "item prologue", "row gather+barrier", "prefetch issue", "LDS reads + MFMA", "vmcnt + commit", "barrier", "epilogue"]
tot = diag.sum(1).mean()
print(f"workgroups {len(diag)}, mean cycles per WG {tot:.0f}")
for k, n in enumerate(names):
    print(f"  {n:22s} {100 * diag[:, k].mean() / tot:6.2f} %   ({diag[:, k].mean():.0f} cycles)")
