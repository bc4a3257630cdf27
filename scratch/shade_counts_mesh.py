"""What happens to the (wave, light) pairs of the benchmark's frame from draw calls (mesh G-buffer, rendered shadow
cubes); library built with -DVKR_EXPERIMENT_COUNT."""
import ctypes, dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vulkanrenderer_amd import synth, passes as P, _lib
from vulkanrenderer_amd.strips import HipBackend, StripPlan
dev = torch.device("cuda:0")
W, H, L, RES = 3840, 2160, 64, 1024
f = synth.make_frame(bench.SEED, W, H, L, RES, 4096)
d = P.DeviceFrame.upload(f, dev)
verts, tris = synth.make_scene_mesh(128)
uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=0.25)
alb, nmap, arm = synth.make_material(np.random.default_rng(bench.SEED + 9), 1024)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
gdraws = [dict(positions=t(verts, np.float32), triangles=t(tris, np.int32), texcoords=t(uv, np.float32), normals=t(nrm, np.float32),
               tangents=t(tan, np.float32), albedo=t(alb, np.uint8), normal=t(nmap, np.uint8), arm=t(arm, np.uint8), model=np.eye(4))]
scene = f.scene.copy()
scene["camera_projection"] = synth.perspective(60.0, W / H, 0.1, 100.0).astype(np.float32).ravel()
env = P.tile_env_cube(P.to_device_f16(synth.make_env_cube(np.random.default_rng(bench.SEED + 10), 512), dev))
be = HipBackend(dev)
plan = StripPlan.make(0, 1, H)
base = dataclasses.replace(d, rows=H, scene=scene, coverage=None, coverage_flags=None, background=None)
fr = be.frame_from_draws(base, gdraws, RES, 4096, plan.total_rows, plan.row0, (env, _lib.ENV_LAYOUT_TILED_4X4))
color = torch.empty((H, W, 4), dtype=torch.float16, device=dev)
thr = torch.empty((H, W, 4), dtype=torch.float16, device=dev)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
torch.cuda.synchronize()
lib.vkr_experiment_shade_counts(buf, 1)
be.shade(fr, color, thr); torch.cuda.synchronize()
lib.vkr_experiment_shade_counts(buf, 1)
c = list(buf)
names = ["no live lane", "(unused)", "one face, taps read", "edge or corner, taps read", "all live lanes fully shadowed (no BRDF)", "BRDF evaluated", "tile class: skip", "tile class: lit"]
tot = c[0] + c[2] + c[3] + c[6] + c[7]
for n, v in zip(names, c): print(f"{n:45s} {v:10d} {v / max(tot, 1):.3f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): be.shade(fr, color, thr)
e1.record(); torch.cuda.synchronize()
print("shade ms", e0.elapsed_time(e1) / 5, "covered", float((fr.albedo[..., 3] != 0).float().mean()))
