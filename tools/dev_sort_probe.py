"""Dev: cost of the occurrence sort inside rbr_review_bag_bwd, eager (rocPRIM radix sort) vs replayed from a hipGraph
(rocPRIM merge sort: no memset nodes), at n_rev x T token positions with a tiny D so the rest of the call is small."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from review_based_recommender_amd import functional as RF
dev = torch.device("cuda:0")
for n_rev, T in ((4096, 64), (8192, 64), (16384, 64)):
    V, D = 50002, 4
    table = torch.randn(V, D, device=dev, requires_grad=True)
    p = torch.arange(1, V, dtype=torch.float64) ** -1.07
    ids = (torch.multinomial(p / p.sum(), n_rev * T, replacement=True) + 1).view(n_rev, T).to(dev)
    mask = torch.ones(n_rev, T, dtype=torch.bool, device=dev)
    def step():
        table.grad = None
        RF.review_bag(table, ids, mask).sum().backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20 * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 20 * 1e6
    print(f"n_pos={n_rev*T}: eager (radix) {eager:.0f} us per fwd+bwd, graph replay (merge sort) {graph:.0f} us", flush=True)
