"""One launch of each IBL kernel at the 2048^2-environment configuration (for counter passes): the precompute as
bench.py times it, then the irradiance pass in the shader's own order for comparison."""
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
foot = P.tile_env_footprints(env)  # the copy the irradiance and prefilter passes read
P.irradiance_by_region_rows(foot, 2048, irr)
P.prefilter_mips(foot, pres, env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=2048)
P.brdf_lut(lut)
tiled = P.tile_env_cube(env)
P.irradiance(tiled, irr, _lib.IRRADIANCE_SEQUENTIAL, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
torch.cuda.synchronize()
