"""Dev: RCCL sanity on one GPU (world_size 1): init, all_reduce with SUM / AVG, barrier, async handles."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.arange(8, dtype=torch.float32, device="cuda")
for op in (dist.ReduceOp.SUM, dist.ReduceOp.AVG, dist.ReduceOp.MAX):
    y = x.clone(); h = dist.all_reduce(y, op=op, async_op=True); h.wait(); torch.cuda.synchronize()
    assert torch.equal(y, x), op
big = torch.ones(15_000_000, device="cuda"); h = dist.all_reduce(big, op=dist.ReduceOp.AVG, async_op=True); h.wait()
dist.barrier(); torch.cuda.synchronize()
print("rccl ok", dist.get_backend(), torch.cuda.get_device_name(0))
dist.destroy_process_group()
