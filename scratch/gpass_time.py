"""Times the geometry pass of the benchmark's mesh at 4K with and without triangle records.  usage: python scratch/gpass_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vulkanrenderer_amd import passes as P, synth
dev = torch.device("cuda:0")
W, H = 3840, 2160
verts, tris = synth.make_scene_mesh(128)
uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=0.25)
alb, nmap, arm = synth.make_material(np.random.default_rng(bench.SEED + 9), 1024)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
gdraws = [dict(positions=t(verts, np.float32), triangles=t(tris, np.int32), texcoords=t(uv, np.float32), normals=t(nrm, np.float32),
               tangents=t(tan, np.float32), albedo=t(alb, np.uint8), normal=t(nmap, np.uint8), arm=t(arm, np.uint8), model=np.eye(4))]
scene = synth.default_scene_info()
scene["camera_projection"] = synth.perspective(60.0, W / H, 0.1, 100.0).astype(np.float32).ravel()
for rec in (True, False, True, False):
    P.gbuffer(gdraws, scene, W, H, dev, triangle_records=rec); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): P.gbuffer(gdraws, scene, W, H, dev, triangle_records=rec)
    e1.record(); torch.cuda.synchronize()
    print("triangle records", rec, "geometry pass ms", round(e0.elapsed_time(e1) / 10, 4))
