"""Timing of the shadow passes at the benchmark configuration (64 lights, 1024^2 cubes, 4096^2 directional map)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vulkanrenderer_amd import passes as P, synth

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
verts, tris = synth.make_scene_mesh(n)
draws = [(torch.from_numpy(verts).to(dev), torch.from_numpy(tris.astype(np.int32)).to(dev), np.eye(4))]
lights = synth.make_lights(np.random.default_rng(bench.SEED), 64)
dl = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(dev)
ls = synth.default_scene_info()["light_space_matrix"]


def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / reps:8.3f} ms", flush=True)


print("mesh:", len(verts), "vertices", len(tris), "triangles")
timed("point shadows 64 x 1024^2, tiled", lambda: P.point_shadow_maps(draws, dl, 64, 1024, tiled=True))
timed("point shadows 64 x 1024^2, row-major", lambda: P.point_shadow_maps(draws, dl, 64, 1024, tiled=False))
timed("point shadows 64 x 256^2 (reference size)", lambda: P.point_shadow_maps(draws, dl, 64, 256, tiled=True))
timed("directional 4096^2", lambda: P.dir_shadow_map(draws, ls, 4096, dev))
cube, _ = P.point_shadow_maps(draws, dl, 64, 1024, tiled=False)
print("covered fraction", float((cube < 1).float().mean()))
