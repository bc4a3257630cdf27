"""RCCL on the box, as far as one GPU allows: a one-rank group, all_reduce, all_gather_into_tensor, gather, barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
t = torch.ones(1024, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = torch.empty(1024, device=dev)
dist.all_gather_into_tensor(out, t)
lst = [torch.empty(1024, device=dev)]
dist.gather(t, lst, dst=0)
dist.barrier()
torch.cuda.synchronize()
print("rccl", torch.cuda.nccl.version(), "ok", float(out.sum()), float(lst[0].sum()))
dist.destroy_process_group()
