"""Times the by-region irradiance order against the sequential one at the bench's sizes and compares the images."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import _lib, passes as P, synth

dev = torch.device("cuda:0")
env_n, n = 2048, 64
env = synth.make_env_cube(np.random.default_rng(5), env_n)
d = P.tile_env_cube(P.to_device_f16(env, dev))
T = _lib.ENV_LAYOUT_TILED_4X4
seq = torch.empty((6, n, n, 4), dtype=torch.float16, device=dev)
reg = torch.empty_like(seq)
P.irradiance(d, seq, 0, env_layout=T)
ws = P.irradiance_by_region(d, reg, env_layout=T)
torch.cuda.synchronize()
print("workspace MB", ws.numel() / 1e6, "uncovered", P.irradiance_uncovered(ws))
a, b = seq.float().cpu().numpy(), reg.float().cpu().numpy()
ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(a), 2.0 ** -14))) - 10)
print("texels that differ", int((P.f16_bits(seq) != P.f16_bits(reg)).any(-1).sum()), "of", 6 * n * n, "max ulps", float((np.abs(a - b) / ulp).max()))
for name, fn in (("sequential", lambda: P.irradiance(d, seq, 0, env_layout=T)), ("by_region", lambda: P.irradiance_by_region(d, reg, env_layout=T, workspace=ws))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "ms", e0.elapsed_time(e1) / 5)
