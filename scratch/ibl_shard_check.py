"""Two gloo ranks sharing one GPU run the sharded IBL precompute at the 2048^2 configuration, with a note per stage."""
import os, sys, datetime, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

def worker(rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29545")
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=60))
    from vulkanrenderer_amd import _lib, passes as P, synth
    from vulkanrenderer_amd import ibl
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    say = lambda m: print(f"[{rank} {time.strftime('%H:%M:%S')}] {m}", flush=True)
    cfg = ibl.IblConfig(env_res=int(os.environ.get("ENV", "2048")), irradiance_res=64, prefilter_res=128, prefilter_mips=5, lut_res=512)
    src = P.to_device_f16(synth.make_equirect(np.random.default_rng(7), 4096, 2048, hdr=True), dev)
    say("source ready")
    pre = ibl.IblPrecompute(cfg, rank, 2, device=dev)
    orig = pre._all_gather_rows
    def traced(flat, per):
        torch.cuda.synchronize(); say(f"all_gather of {tuple(flat.shape)} per {per}")
        orig(flat, per)
        torch.cuda.synchronize(); say("  done")
    pre._all_gather_rows = traced
    pre.run(src, _lib.SRC_RGBA16F)
    torch.cuda.synchronize(); say("run complete")
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, nprocs=2, join=True)
