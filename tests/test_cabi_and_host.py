"""CPU-only checks of the drop-in boundary: the shared library loads and exports every symbol that
include/rbr_hip.h declares, the ctypes table mirrors the header, plan queries work without a GPU, and
the host-side modules keep the reference's names / keys / error behaviour.  No kernel is launched."""
import ctypes as C
import os
import re

import pytest
import torch

import synth
from helpers import quiet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "rbr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rbr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from review_based_recommender_amd import _lib
    names = _header_functions()
    assert len(names) >= 25
    handle = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"librbr_hip.so lacks {missing}"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    L = _lib.lib()
    assert L.rbr_version() >= 1


def test_plan_queries_and_error_reporting_without_gpu():
    from review_based_recommender_amd import _lib
    L = _lib.lib()
    d = _lib.make_desc(512, 512, 300, 50002, [3, 5, 7], [50, 50, 50], _lib.PAD_SAME, _lib.ACT_RELU, 0)
    # 5 channel tiles x 7 taps x 5 chunks x 32 slots x 60 floats
    assert L.rbr_textcnn_packed_floats(C.byref(d)) == 5 * 7 * 5 * 32 * 60
    assert L.rbr_textcnn_partial_elems(C.byref(d)) == 512 * 16 * 160 + 2 * 512 * 16 + 64
    assert L.rbr_textcnn_bwd_ws_floats(C.byref(d)) > 0
    bad = _lib.make_desc(4, 16, 8, 20, [4], [6], _lib.PAD_SAME, _lib.ACT_RELU, 0)     # even 'same' width
    assert L.rbr_textcnn_packed_floats(C.byref(bad)) == 0
    assert b"odd" in L.rbr_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(-1, "demo")


def test_product_refuses_cpu_tensors():
    from review_based_recommender_amd import functional as RF
    cfg = synth.DEEPCONN_CFGS["tiny"]
    p = synth.deepconn_params(cfg, 0)
    b = synth.deepconn_batch(cfg, 1)
    with pytest.raises(RuntimeError, match="HIP device"):
        RF.textcnn(p["word_embeddings.embedding.weight"], b["u_docs"], b["u_masks"],
                   [p["ngram.feature_layer.0.list_of_conv1d.0.weight"]], [p["ngram.feature_layer.0.list_of_conv1d.0.bias"]])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "review-based-recommender_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


@pytest.mark.parametrize("which", ["deepconn", "narre", "datt"])
def test_state_dict_keys_and_init(which):
    from review_based_recommender_amd.models.deepconn.deepconn import DeepCoNNpp
    from review_based_recommender_amd.models.dual_att.dual_att import DualAtt
    from review_based_recommender_amd.models.narre.narre import NARRE
    if which == "deepconn":
        c = synth.DEEPCONN_CFGS["tiny"]
        m = quiet(DeepCoNNpp, c["U"], c["I"], c["V"], c["kz"], c["D"], c["H"], c["K"], c["L"], None, 0.5)
        ref = synth.deepconn_params(c, 0)
        assert float(m.word_embeddings.embedding.weight[0].abs().max()) == 0.0   # pad row zero at init
        assert float(m.user_feat.ebd.weight[0].abs().max()) > 0.0                 # quirk 4: re-initialised non-zero
        assert torch.all(m.user_feat.b == 0.1) and float(m.fm.g_bias) == pytest.approx(0.1)
    elif which == "narre":
        c = synth.NARRE_CFGS["tiny"]
        m = quiet(NARRE, c["U"], c["I"], c["V"], c["kz"], c["H"], c["D"], c["A"], c["K"], c["R"], c["T"], 0.5, 0, 0, 0,
                  None, "CNN")
        ref = synth.narre_params(c, 0)
        assert m.user_att.ebd_vals.weight.shape[0] == c["I"]      # user tower keyed by ITEM ids
    else:
        c = synth.DATT_CFGS["tiny"]
        m = quiet(DualAtt, c["V"], c["L"], c["win"], c["l_out"], c["g_out"], c["E"], c["h1"], c["h2"], 0.5, None)
        ref = synth.datt_params(c, 0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    m.load_state_dict(ref)   # reference checkpoints load unchanged


def test_pretrained_embeddings_and_freeze():
    from review_based_recommender_amd.models.deepconn.layers import WordEmbedding
    w = torch.arange(12, dtype=torch.float32).view(4, 3) + 1.0
    e = quiet(WordEmbedding, 4, 3, pretrained_embeddings=w.numpy(), freeze_embeddings=True)
    assert torch.equal(e.embedding.weight.detach(), w)       # pad row NOT zeroed when pretrained rows are loaded
    assert not e.embedding.weight.requires_grad


def test_stack_rows_and_clone_adjacent_host_logic():
    """functional.stack_rows / clone_adjacent are pure tensor bookkeeping: the same on CPU tensors.  Adjacent halves stack as a
    view; anything else falls back to torch.cat."""
    import torch

    from review_based_recommender_amd import functional as RF
    g = torch.Generator().manual_seed(0)
    a, b = torch.randint(0, 50, (4, 6), generator=g), torch.randint(0, 50, (4, 6), generator=g)
    m1, m2 = torch.rand(4, 6, generator=g) > 0.5, torch.rand(4, 6, generator=g) > 0.5
    ids = torch.randint(0, 9, (4,), generator=g)
    ca, cb, cm1, cm2, cids = RF.clone_adjacent((a, b, m1, m2, ids))
    for x, y in ((ca, a), (cb, b), (cm1, m1), (cm2, m2), (cids, ids)):
        assert torch.equal(x, y) and x.data_ptr() != y.data_ptr()
    st = RF.stack_rows(ca, cb)
    assert st.data_ptr() == ca.data_ptr() and st.shape == (8, 6) and torch.equal(st, torch.cat([a, b]))
    assert RF.stack_rows(cm1, cm2).data_ptr() == cm1.data_ptr()
    assert torch.equal(RF.stack_rows(a, b), torch.cat([a, b])) and RF.stack_rows(a, b).data_ptr() != a.data_ptr()
    assert RF.stack_rows(cb, ca).data_ptr() != cb.data_ptr()                     # wrong order
    assert RF.stack_rows(ca[:, :3], cb[:, :3]).data_ptr() != ca.data_ptr()        # non-contiguous halves
    # NARRE's [bz, R, T] -> [bz*R, T] views keep the adjacency
    u, i = RF.clone_adjacent((torch.arange(24).view(2, 3, 4), torch.arange(24, 48).view(2, 3, 4)))
    st = RF.stack_rows(u.reshape(-1, 4), i.reshape(-1, 4))
    assert st.data_ptr() == u.data_ptr() and torch.equal(st, torch.arange(48).view(12, 4))


def test_step_input_block_layout_host_logic():
    """train_step._flat_layout / _flat_views: all inputs of a step in one block, tower pairs back to back (stack_rows sees
    a view), everything else 256-byte aligned; a packed copy of another batch lands field by field."""
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.train_step import _flat_layout, _flat_views
    g = torch.Generator().manual_seed(1)
    u, i = torch.randint(0, 50, (5, 7), generator=g), torch.randint(0, 50, (5, 7), generator=g)
    mu, mi = u != 0, i != 0
    uid, iid = torch.randint(0, 9, (5,), generator=g), torch.randint(0, 9, (5,), generator=g)
    r = torch.rand(5, generator=g)
    like = [u, i, mu, mi, uid, iid, r]
    lay = _flat_layout(like)
    assert len(lay) == len(like) + 1 and lay[0] == 0 and lay[1] == u.numel() * 8          # the pair is contiguous
    assert all(lay[k] % 256 == 0 for k in (0, 2, 4, 6, 7))
    flat = torch.zeros(lay[-1], dtype=torch.uint8)
    views = _flat_views(flat, lay, like)
    for v, src in zip(views, like):
        v.copy_(src)
    for v, src in zip(views, like):
        assert v.dtype == src.dtype and torch.equal(v, src)
    assert RF.stack_rows(views[0], views[1]).data_ptr() == views[0].data_ptr()
    assert RF.stack_rows(views[2], views[3]).data_ptr() == views[2].data_ptr()
    assert torch.equal(RF.stack_rows(views[0], views[1]), torch.cat([u, i]))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a plain invocation launches two ranks through torch.distributed.run before touching the
    GPU.  Without a GPU every rank stops at the device check -- which is the proof that both were started."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the real multi-rank run is the driver's")
    assert r.returncode != 0
    assert out.count("bench.py needs an MI355X") >= 2, out[-2000:]


def test_product_library_holds_no_wrong_result_switches():
    """the RBR_DEV_* ablation / tuning switches (no atomics, no accumulation phase, extra LDS, generic GEMM)
    exist only in -DRBR_DIAG builds (RBR_DIAG=1 python build.py); the product library must not even contain their names, so no
    environment variable can turn gradient accumulation off in it."""
    import os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "review-based-recommender_amd", "csrc", "librbr_hip.so")
    blob = open(lib, "rb").read()
    for name in (b"RBR_DEV_DX_ABLATE", b"RBR_DEV_DX_WIN", b"RBR_DEV_CONV_EXTRA_LDS", b"RBR_DEV_GENERIC_GEMM"):
        assert name not in blob, name


def test_row_gradient_hand_off_is_scoped_and_weak():
    """the compact-row-gradient registry must not (a) keep an optimizer alive, (b) answer for an optimizer that
    merely exists.  functional keeps a weak reference and asks wants_row_grad() per backward; HipClipAdam says yes only inside
    its row_grad_scope() (opened by train_step() / GraphedTrainStep around the forward + backward it will finish)."""
    import gc
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.train_step import HipClipAdam

    class Sink:
        def __init__(self):
            self.armed = False

        def wants_row_grad(self, table):
            return self.armed

    table = torch.zeros(8, 4)
    s = Sink()
    RF.set_row_grad_sink(table, s)
    assert RF._row_grad_sink_for(table) is None            # registered, not armed: the backward stays dense
    s.armed = True
    assert RF._row_grad_sink_for(table) is s
    del s
    gc.collect()
    assert RF._row_grad_sink_for(table) is None            # the registry did not keep it alive ...
    assert table.data_ptr() not in RF._ROW_GRAD_SINKS      # ... and forgot the dead entry

    p = torch.nn.Parameter(torch.zeros(HipClipAdam.ROW_GRAD_MIN_ROWS, 4))
    opt = HipClipAdam([p])                                  # a CPU parameter registers nothing, but the scope logic is host-only
    opt._row_tables.append(p)
    assert not opt.wants_row_grad(p)
    with opt.row_grad_scope():
        assert opt.wants_row_grad(p)
        with opt.row_grad_scope():
            assert opt.wants_row_grad(p)
        assert opt.wants_row_grad(p)
    assert not opt.wants_row_grad(p)
