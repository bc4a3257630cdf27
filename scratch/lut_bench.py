"""BRDF LUT 512^2: time per launch and bit-equality with the oracle (VKR_HIP_LIB selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanrenderer_amd import passes as P
from tests import oracle
dev = "cuda:0"
lut = torch.empty((512, 512, 2), dtype=torch.float16, device=dev)
P.brdf_lut(lut); torch.cuda.synchronize()
ok = np.array_equal(P.f16_bits(lut), oracle.brdf_lut(512))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): P.brdf_lut(lut)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(os.environ.get("VKR_HIP_LIB", "default"), f"{ms:.4f} ms  {60*1024*512*512/ms/1e9:.1f} TFLOP/s canonical  bit-exact={ok}", flush=True)
