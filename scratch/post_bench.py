"""Times the post-chain kernels alone (HIP events): a 4K frame, whose images fit the Infinity Cache, and a
16384 x 2048 strip, whose images (671 MB per launch) do not, so every launch reads from HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanrenderer_amd import _lib, passes as P
dev = torch.device("cuda", 0)


def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(W, H, bands=(0,)):
    thr = (torch.rand((H, W, 4), device=dev) * 2).half()
    hdr = (torch.rand((H, W, 4), device=dev) * 4).half()
    b1 = torch.empty_like(thr); b2 = torch.empty_like(thr)
    ldr = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    px = W * H
    res = {}
    res["bloom_h"] = (timeit(lambda: P.bloom_blur(thr, b1, True, _lib.WRAP_REPEAT)), 16)
    res["bloomv_post"] = (timeit(lambda: P.bloomv_postprocess(b1, hdr, ldr)), 20)
    res["one_kernel"] = (timeit(lambda: P.bloom_postprocess(thr, hdr, ldr)), 20)
    ref = torch.empty_like(ldr)
    P.bloomv_postprocess(b1, hdr, ref)
    same = bool(torch.equal(ref, ldr))
    print(f"{W}x{H}", os.environ.get("VKR_HIP_LIB", "default"), "band", os.environ.get("VKR_BF_BAND", "auto"),
          " ".join(f"{k}={v*1e3:.1f}us({b*px/v/1e6:.0f}GB/s)" for k, (v, b) in res.items()), "equal" if same else "DIFFERENT", flush=True)


run(3840, 2160)
run(16384, 2048)
