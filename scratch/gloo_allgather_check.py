"""Two gloo ranks sharing one GPU: does all_gather_into_tensor take CUDA tensors (in place, input = a slice of the output)?"""
import os, sys, datetime
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

def worker(rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=30))
    dev = torch.device("cuda", 0)
    flat = torch.zeros((8, 4), dtype=torch.float16, device=dev)
    flat[rank * 4:(rank + 1) * 4] = rank + 1
    try:
        dist.all_gather_into_tensor(flat, flat[rank * 4:(rank + 1) * 4])
        torch.cuda.synchronize()
        print(rank, "in place on CUDA:", flat[:, 0].tolist(), flush=True)
    except Exception as e:
        print(rank, "in place on CUDA failed:", type(e).__name__, str(e)[:200], flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, nprocs=2, join=True)
