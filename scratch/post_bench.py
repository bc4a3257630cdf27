"""Times the post-chain kernels alone on a 4K frame (HIP events, 200 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanrenderer_amd import _lib, passes as P
dev = torch.device("cuda", 0)
W, H = 3840, 2160
thr = (torch.rand((H, W, 4), device=dev) * 2).half()
hdr = (torch.rand((H, W, 4), device=dev) * 4).half()
b1 = torch.empty_like(thr); b2 = torch.empty_like(thr)
ldr = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
px = W * H
res = {}
res["bloom_h"] = timeit(lambda: P.bloom_blur(thr, b1, True, _lib.WRAP_REPEAT))
res["bloom_v"] = timeit(lambda: P.bloom_blur(b1, b2, False, _lib.WRAP_CLAMP))
res["post"] = timeit(lambda: P.postprocess(hdr, b2, ldr))
res["bloomv_post"] = timeit(lambda: P.bloomv_postprocess(b1, hdr, ldr))
bytes_ = {"bloom_h": 16, "bloom_v": 16, "post": 20, "bloomv_post": 20}
print(os.environ.get("VKR_HIP_LIB", "default"), " ".join(f"{k}={v*1e3:.1f}us({bytes_[k]*px/v/1e6:.0f}GB/s)" for k, v in res.items()))
