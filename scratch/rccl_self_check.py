"""One rank over RCCL on one GPU: the process-group calls the strip renderer makes (init with device_id,
barrier, all_reduce, batch_isend_irecv on row slices of a larger tensor -- here to and from the rank itself --
on the main and on a side stream).  The development box has one GPU, so this is as much of the RCCL
transport as can be run there; the N > 1 path proper runs in the driver's scaling bench."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("rccl", torch.cuda.nccl.version())
dist.barrier()
t = torch.ones(4, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
img = torch.arange(16 * 64 * 4, device=dev, dtype=torch.float32).reshape(16, 64, 4).to(torch.float16)
want = img.clone()
ops = [dist.P2POp(dist.isend, img[4:8], 0), dist.P2POp(dist.irecv, img[0:4], 0),
       dist.P2POp(dist.isend, img[8:12], 0), dist.P2POp(dist.irecv, img[12:16], 0)]
for req in dist.batch_isend_irecv(ops):
    req.wait()
torch.cuda.synchronize()
assert torch.equal(img[0:4], want[4:8]) and torch.equal(img[12:16], want[8:12]), "halo rows did not arrive"
side = torch.cuda.Stream(device=dev)
ldr = torch.full((8, 64, 4), 7, dtype=torch.uint8, device=dev)
got = torch.zeros_like(ldr)
ev = torch.cuda.Event(); ev.record()
with torch.cuda.stream(side):
    side.wait_event(ev)
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, ldr, 0), dist.P2POp(dist.irecv, got, 0)]):
        req.wait()
torch.cuda.synchronize()
assert torch.equal(got, ldr)
dist.destroy_process_group()
print("rccl self check ok")
