"""Per-launch timing of the IBL kernels at the 2048^2-environment configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vulkanrenderer_amd import _lib, passes as P, synth

dev = "cuda:0"
rng = np.random.default_rng(7)
src = P.to_device_f16(synth.make_equirect(rng, 4096, 2048, hdr=True), dev)
env = torch.empty((6, 2048, 2048, 4), dtype=torch.float16, device=dev)
irr = torch.empty((6, 64, 64, 4), dtype=torch.float16, device=dev)
pres = [torch.empty((6, 128 >> m, 128 >> m, 4), dtype=torch.float16, device=dev) for m in range(5)]
lut = torch.empty((512, 512, 2), dtype=torch.float16, device=dev)
P.equirect_to_cube(src, _lib.SRC_RGBA16F, env)


def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / reps:8.3f} ms", flush=True)


timed("equirect", lambda: P.equirect_to_cube(src, _lib.SRC_RGBA16F, env))
timed("irradiance sequential", lambda: P.irradiance(env, irr, _lib.IRRADIANCE_SEQUENTIAL), 3)
timed("irradiance phi slices", lambda: P.irradiance(env, irr, _lib.IRRADIANCE_PHI_SLICES), 3)
for m in range(5):
    timed(f"prefilter mip {m} r={m / 4}", lambda m=m: P.prefilter(env, pres[m], m / 4.0))
timed("tile env", lambda: P.tile_env_cube(env))
tiled = P.tile_env_cube(env)
T = _lib.ENV_LAYOUT_TILED_4X4
timed("irradiance sequential, tiled", lambda: P.irradiance(tiled, irr, _lib.IRRADIANCE_SEQUENTIAL, env_layout=T), 3)
for m in range(5):
    timed(f"prefilter mip {m}, tiled", lambda m=m: P.prefilter(tiled, pres[m], m / 4.0, env_layout=T))
timed("brdf lut", lambda: P.brdf_lut(lut))
