"""Sweep of environment and cube sizes through the by-region irradiance kernel: every texel must account for all of its
samples, and the image must stay within one binary16 ulp of the kernel in the shader's order."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import _lib, passes as P

dev = torch.device("cuda:0")
bad = 0
for env_n in (1, 2, 7, 64, 100, 512, 516, 1024, 1100, 1536, 2048, 2560, 4096, 8192):
    env = (torch.rand((6, env_n, env_n, 4), device=dev) * 4.0).to(torch.float16)
    layout = _lib.ENV_LAYOUT_LINEAR
    if env_n % 4 == 0:
        env, layout = P.tile_env_cube(env), _lib.ENV_LAYOUT_TILED_4X4
    for n in (1, 2, 3, 5, 8, 16, 33, 64) if env_n < 8192 else (2, 16):
        out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=dev)
        seq = torch.zeros_like(out)
        ws = P.irradiance_by_region(env, out, env_layout=layout)
        P.irradiance(env, seq, 0, env_layout=layout)
        unc = P.irradiance_uncovered(ws)
        a, b = seq.float().cpu().numpy(), out.float().cpu().numpy()
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(a), 2.0 ** -14))) - 10)
        nan_same = np.array_equal(np.isnan(a), np.isnan(b))
        worst = float(np.nanmax(np.abs(a - b) / ulp)) if np.isfinite(a).any() else 0.0
        flag = "" if unc == 0 and nan_same and worst <= 1.0 else "   <-- BAD"
        bad += bool(flag)
        print(f"env {env_n:5d} cube {n:3d}: uncovered {unc}, NaN texels agree {nan_same}, worst difference {worst:.2f} ulp{flag}", flush=True)
    del env
print("bad cases:", bad)
