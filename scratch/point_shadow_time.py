"""Times the point-shadow pass of the benchmark's scene (64 lights, 1024^2 cubes).  usage: python scratch/point_shadow_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vulkanrenderer_amd import passes as P, synth
device = torch.device("cuda:0")
verts, tris = synth.make_scene_mesh(128)
draws = [(torch.from_numpy(verts).to(device), torch.from_numpy(tris.astype(np.int32)).to(device), np.eye(4))]
lights_np = synth.make_lights(np.random.default_rng(bench.SEED), 64)
lights = torch.from_numpy(np.frombuffer(lights_np.tobytes(), dtype=np.uint8).copy()).to(device)
def run():
    return P.point_shadow_maps(draws, lights, 64, 1024)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
cubes, layout = run()
print("point shadow pass ms", round(e0.elapsed_time(e1) / 10, 4), "covered", float((cubes < 25.0).float().mean()))
