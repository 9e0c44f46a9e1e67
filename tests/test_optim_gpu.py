"""HipClipAdam (csrc/clip_adam.hip) against torch's clip_grad_norm_ + Adam on the same parameters and gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(5003, 37), (64, 300, 3), (1,), (7,), (130,), (4096,), (33, 5)]
    return [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]


@pytest.mark.parametrize("max_norm", [5.0, 1e6, None])
def test_clip_adam_matches_torch(max_norm):
    from review_based_recommender_amd.train_step import HipClipAdam
    dev = torch.device("cuda", 0)
    pa, pb = _params(dev, 0), _params(dev, 0)
    oa = torch.optim.Adam(pa, lr=2e-3)
    ob = HipClipAdam(pb, lr=2e-3)
    gen = torch.Generator().manual_seed(1)
    for step in range(5):
        # gradients: dense noise, a mostly-zero "embedding" gradient, and one view at an odd offset of a flat buffer
        flat = torch.randn(7 + 130 + 3, generator=gen).to(dev)
        for k, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=gen).to(dev) * (10.0 if step == 2 else 0.1)
            if k == 0:
                gr[torch.rand(a.shape[0], generator=gen).to(dev) < 0.7] = 0
            if k == 3:
                gr = flat[3:10].view_as(a)          # 4-byte aligned only
            a.grad, b.grad = gr.clone(), gr.clone() if k != 3 else flat.clone()[3:10].view_as(b)
        if max_norm is not None:
            ref_norm = torch.nn.utils.clip_grad_norm_(pa, max_norm)
        else:
            ref_norm = torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in pa]))
        oa.step()
        got_norm = ob.clip_and_step(max_norm)
        torch.cuda.synchronize()
        assert abs(float(got_norm) - float(ref_norm)) <= 2e-6 * float(ref_norm)
        for a, b in zip(pa, pb):
            assert torch.allclose(a.grad, b.grad, rtol=2e-6, atol=1e-9)          # clipped gradients are left behind
            # rounding-level gradient differences can flip +-lr on near-zero gradients: compare with a small absolute slack
            assert float((a - b).abs().max()) <= 2e-6 + 1e-3 * 2e-3
        assert float(ob.state[pb[0]]["step"]) == step + 1


def test_state_dict_interchanges_with_torch_adam():
    from review_based_recommender_amd.train_step import HipClipAdam
    dev = torch.device("cuda", 0)
    pa, pb = _params(dev, 3), _params(dev, 3)
    oa = torch.optim.Adam(pa, lr=2e-3)
    for p in pa:
        p.grad = torch.ones_like(p)
    oa.step()
    ob = HipClipAdam(pb, lr=2e-3)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))      # load_state_dict keeps same-device tensors by reference
    for a, b in zip(pa, pb):
        b.data.copy_(a.data)
        a.grad = torch.full_like(a, 0.5)
        b.grad = torch.full_like(b, 0.5)
    oa.step()
    ob.clip_and_step(None)
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    sd = ob.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}


def test_hip_clip_adam_with_more_than_64_tensors():
    """clipping is no longer refused beyond RBR_OPT_MAX_TENSORS -- the norm / scaling then come from
    clip_grad_norm_ and the Adam launches run per 64 tensors; the result equals torch's clip + Adam."""
    from review_based_recommender_amd.train_step import HipClipAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 5)] * 40 + [(33,)] * 40 + [(1,)] * 10
    ps_a = [torch.nn.Parameter(torch.randn(*s, generator=g).to("cuda:0")) for s in shapes]
    ps_b = [torch.nn.Parameter(p.detach().clone()) for p in ps_a]
    grads = [torch.randn(*s, generator=g).to("cuda:0") * 3 for s in shapes]
    oa, ob = HipClipAdam(ps_a, lr=2e-3), torch.optim.Adam(ps_b, lr=2e-3)
    for step in range(3):
        for pa, pb, gr in zip(ps_a, ps_b, grads):
            pa.grad, pb.grad = gr.clone() * (step + 1), gr.clone() * (step + 1)
        na = oa.clip_and_step(5.0)
        nb = torch.nn.utils.clip_grad_norm_(ps_b, 5.0)
        ob.step()
        assert abs(float(na) - float(nb)) <= 1e-5 * float(nb)
    for pa, pb in zip(ps_a, ps_b):
        assert float((pa - pb).abs().max()) <= 2e-6


@pytest.mark.parametrize("max_norm", [None, 0.5])
@pytest.mark.parametrize("D", [300, 20])
def test_exchanged_rows_in_owner_layout_match_the_dense_gradient(max_norm, D):
    """HipClipAdam.put_exchanged_rows: the averaged table gradient in the layout the owner-partitioned exchange leaves it in
    (distributed.TapExchange: N slabs of ceil(V/N) + 1 rows, token t at [t % N][t // N], the extra row of slab r carrying the
    sum of squares of that slab) read through the static token -> row map, against the same gradient as a dense .grad:
    the same parameters and Adam state bit for bit without clipping (D = 300: the wave-per-row path of the kernel, D = 20:
    its generic path), to rounding with it (the norm is summed from N partials instead of over the dense tensor)."""
    from review_based_recommender_amd import functional as RF
    from review_based_recommender_amd.train_step import HipClipAdam
    dev = torch.device("cuda", 0)
    V, N = 5003, 3
    gen = torch.Generator().manual_seed(5)
    table0 = torch.randn(V, D, generator=gen).to(dev)
    other0 = torch.randn(130, generator=gen).to(dev)
    v_own = (V + N - 1) // N
    t = torch.arange(V, device=dev)
    row_map = ((t % N) * (v_own + 1) + t // N).to(torch.int32)
    models = []
    for _ in range(2):
        ps = [table0.clone().requires_grad_(True), other0.clone().requires_grad_(True)]
        models.append((ps, HipClipAdam(ps, lr=2e-3, row_grads=False)))
    for step in range(3):
        g = torch.randn(V, D, generator=gen).to(dev) * 0.1
        g[torch.rand(V, generator=gen).to(dev) < 0.3] = 0
        g_other = torch.randn(130, generator=gen).to(dev)
        (pd, od), (pr, orr) = models
        pd[0].grad, pd[1].grad = g.clone(), g_other.clone()
        slabs = torch.zeros(N, v_own + 1, D, device=dev)
        for r in range(N):
            mine = g[r::N]
            slabs[r, :mine.shape[0]] = mine
            slabs[r, v_own, 0] = torch.linalg.vector_norm(slabs[r, :v_own]).square()
        sq = slabs[:, v_own, 0].contiguous()
        rg = RF.RowGradient(pr[0], slabs.view(-1, D), sq, row_map.data_ptr(), (row_map, slabs))
        assert orr.put_exchanged_rows(pr[0], rg)
        assert pr[0].grad is None
        pr[1].grad = g_other.clone()
        nd = od.clip_and_step(max_norm)
        nr = orr.clip_and_step(max_norm)
        torch.cuda.synchronize()
        assert abs(float(nd) - float(nr)) <= 1e-5 * float(nd)
        for a, b in zip(pd, pr):
            if max_norm is None:
                assert torch.equal(a, b)
                assert torch.equal(od.state[a]["exp_avg"], orr.state[b]["exp_avg"])
                assert torch.equal(od.state[a]["exp_avg_sq"], orr.state[b]["exp_avg_sq"])
            else:
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
        orr.zero_grad(); od.zero_grad()
    assert float(torch.equal(rg.to_dense(), g) if max_norm is None else True)
