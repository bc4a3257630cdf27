"""Times the irradiance pass at 2048^2 -> 64^2 in the by-region order with row sums against the other orders."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanrenderer_amd import _lib, passes as P, synth
dev = torch.device("cuda:0")
env_n, n = int(os.environ.get("ENV_N", 2048)), int(os.environ.get("CUBE_N", 64))
src = torch.from_numpy(synth.make_equirect(np.random.default_rng(5), 4096, 2048, hdr=True).view(np.float16)).to(dev)
env = torch.empty((6, env_n, env_n, 4), dtype=torch.float16, device=dev)
P.equirect_to_cube(src, _lib.SRC_RGBA16F, env)
T4 = P.tile_env_cube(env)
out_rows = torch.zeros((6, n, n, 4), dtype=torch.float16, device=dev)
out_reg = torch.zeros_like(out_rows); out_seq = torch.zeros_like(out_rows)
tiles = P.tile_env_footprints(env)
ws = P.irradiance_by_region_rows(tiles, env_n, out_rows)
print("uncovered", P.irradiance_uncovered(ws), "workspace MB", ws.numel() / 1e6, "tiles MB", tiles.numel() / 1e6)
ws2 = P.irradiance_by_region(T4, out_reg, env_layout=1)
P.irradiance(T4, out_seq, 0, env_layout=1)
a, b, c = out_seq.float().cpu().numpy(), out_rows.float().cpu().numpy(), out_reg.float().cpu().numpy()
ulp = np.maximum(np.abs(a), 2.0 ** -14); ulp = 2.0 ** (np.floor(np.log2(ulp)) - 10)
print("rows vs sequential: texels that differ", int((a != b).any(-1).sum()), "max ulps", float((np.abs(a - b) / ulp).max()))
print("region vs sequential: texels that differ", int((a != c).any(-1).sum()), "max ulps", float((np.abs(a - c) / ulp).max()))
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("footprint tiles  ms", timed(lambda: P.tile_env_footprints(env)))
print("4x4 tiles        ms", timed(lambda: P.tile_env_cube(env)))
print("rows order       ms", timed(lambda: P.irradiance_by_region_rows(tiles, env_n, out_rows, workspace=ws)))
print("region order     ms", timed(lambda: P.irradiance_by_region(T4, out_reg, env_layout=1, workspace=ws2)))
