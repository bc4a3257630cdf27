"""One-kernel bloom chain alone on the 16384 x 2048 strip (for counter passes): 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanrenderer_amd import _lib, passes as P
dev = torch.device("cuda", 0)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 2048)
thr = (torch.rand((H, W, 4), device=dev) * 2).half()
hdr = (torch.rand((H, W, 4), device=dev) * 4).half()
ldr = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
for _ in range(20):
    P.bloom_postprocess(thr, hdr, ldr)
torch.cuda.synchronize()
