"""Times the equirect -> cube pass at 4096x2048 -> 2048^2 (both source formats) and the two re-layouts of the cube."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanrenderer_amd import _lib, passes as P, synth
dev = "cuda:0"
rng = np.random.default_rng(7)
src = P.to_device_f16(synth.make_equirect(rng, 4096, 2048, hdr=True), dev)
src8 = torch.from_numpy(synth.make_equirect(rng, 4096, 2048, hdr=False)).to(dev)
env = torch.empty((6, 2048, 2048, 4), dtype=torch.float16, device=dev)
def timed(name, fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / reps:8.4f} ms", flush=True)
timed("equirect RGBA16F source", lambda: P.equirect_to_cube(src, _lib.SRC_RGBA16F, env))
timed("equirect RGBA8 sRGB source", lambda: P.equirect_to_cube(src8, _lib.SRC_RGBA8_SRGB, env))
timed("4x4 tiles", lambda: P.tile_env_cube(env))
timed("footprint tiles", lambda: P.tile_env_footprints(env))
