"""Soak of the one-process strip contexts: three renderer objects share the GPU and swap halo rows with
vkr_strip_exchange for 200 frames from draw calls without a host-side wait in between; every 20th frame is read
back and compared with the same frame from one renderer object."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanrenderer_amd import _lib, synth
from vulkanrenderer_amd.renderer import HipRenderer, strip_exchange, strip_plan
from tests.test_gpu_parity import _gpass_scene, _camera_scene

w, h, nl, n_strips = 640, 360, 6, 3
draws = _gpass_scene(w, h)
lights = [synth.make_lights(np.random.default_rng(6 + k), nl) for k in range(5)]
sphere = synth.reference_light_volume()
env = synth.make_env_cube(np.random.default_rng(2), 16)
kw = dict(max_lights=8, dir_shadow_res=128, point_shadow_res=64, env_res=16, irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32)
plans = [strip_plan(h, n_strips, k) for k in range(n_strips)]
whole = HipRenderer(w, h, **kw)
strips = [HipRenderer(w, rows, frame_rows=h, row0=row0, **kw) for row0, rows in plans]
for r in [whole] + strips:
    r.UploadEnvironmentCubemap(env)
    r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
    r.SetLightVolumeMesh(*sphere)
    r.handles = []
    for d in draws:
        mesh = r.CreateMesh(d["positions"], d["triangles"], d["texcoords"], d["normals"], d["tangents"])
        r.handles.append((mesh, r.CreateMaterial(d["albedo"], d["normal"], d["arm"]), d["model"]))

def begin(r, k):
    scene = _camera_scene((2.0 + 0.01 * (k % 50), 1.6, -4.0), aspect=w / h)
    scene["dir_light_color"] = (40.0, 36.0, 30.0)
    r.UpdateSceneInfo(scene)
    r.SetLightInfos(lights[k % 5], _lib.COVERAGE_MESH if k % 3 else _lib.COVERAGE_FULLSCREEN)
    r.BeginFrame()
    for mesh, material, model in r.handles:
        r.DrawModel(mesh, model, material)

bad = 0
for k in range(200):
    check = k % 20 == 19
    slot = whole.CurrentFrame
    if check:
        begin(whole, k); whole._call(whole._lib.vkr_end_frame)
    else:   # keep the single renderer's frame counter in step
        begin(whole, k); whole._call(whole._lib.vkr_end_frame)
    for r in strips:
        begin(r, k)
    for r in strips:
        r._call(r._lib.vkr_strip_shade)
    for up, lo in zip(strips, strips[1:]):
        strip_exchange(up, lo)
    for r in strips:
        r.StripFinish()
    if check:
        got = np.concatenate([r.Read(_lib.TARGET_LDR, slot)[:rows] for r, (_, rows) in zip(strips, plans)], 0)
        same = np.array_equal(got, whole.Read(_lib.TARGET_LDR, slot))
        bad += int(not same)
print(f"200 frames over {n_strips} strip contexts, {bad} of 10 checked frames differ from one context", flush=True)
sys.exit(1 if bad else 0)
