"""Time of the mip-0 prefilter launch (roughness 0, 128^2 faces from a 2048^2 environment in footprint tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanrenderer_amd import _lib, passes as P, synth
dev = "cuda:0"
env = P.to_device_f16(synth.make_env_cube(np.random.default_rng(7), 2048), dev)
foot = P.tile_env_footprints(env)
out = torch.empty((6, 128, 128, 4), dtype=torch.float16, device=dev)
for name, rough in (("roughness 0", 0.0), ("roughness 0.25", 0.25)):
    for _ in range(2): P.prefilter(foot, out, rough, env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=2048)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): P.prefilter(foot, out, rough, env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=2048)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('VKR_HIP_LIB', 'default'):40s} {name}: {e0.elapsed_time(e1) / 10:.4f} ms")
