"""Times the shading kernel alone on the benchmark frame (HIP events): without masks, with the masks and per-light
words of the coverage pass (reference radii: every light covers every pixel), and with the masks alone.
VKR_HIP_LIB selects a library build; VKR_LIGHTS the number of lights; VKR_ONLY=plain|flags|masks runs one form
(for counter passes)."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vulkanrenderer_amd import passes as P, synth
dev = torch.device("cuda", 0)
L = int(os.environ.get("VKR_LIGHTS", "64"))
bench.HEIGHT = int(os.environ.get("VKR_HEIGHT", bench.HEIGHT))  # e.g. 270: the strip of an 8-way split
scene = bench.build_scene(dev, L, bench.CUBE_RES, bench.DIR_RES)
if os.environ.get("VKR_C5"):
    # the per-GPU share of the 16384^2 / 256-light configuration, built as bench.py builds it: a 4096-wide synthetic
    # strip repeated x4, the L lights and their depth cubes repeated to 256
    bench.C5_WIDTH = int(os.environ.get("VKR_C5_WIDTH", bench.C5_WIDTH))      # experiments: other multiples of 4096
    bench.C5_LIGHTS = int(os.environ.get("VKR_C5_LIGHTS", bench.C5_LIGHTS))   # and multiples of L
    reps = bench.C5_LIGHTS // L
    g = synth.make_gbuffer(np.random.default_rng(bench.SEED + 2000), 4096, bench.C5_ROWS)
    wide = [np.ascontiguousarray(np.tile(a, (1, bench.C5_WIDTH // 4096, 1))) for a in g]
    base = P.DeviceFrame.upload(bench.frame_of(scene, [a[:8] for a in g], 4096, 8), dev)
    lights = np.concatenate([scene["lights"]] * reps)
    d = dataclasses.replace(base, width=bench.C5_WIDTH, rows=bench.C5_ROWS, n_lights=bench.C5_LIGHTS,
                            lights=torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(dev),
                            cube_shadow=base.cube_shadow.repeat(reps, 1, 1, 1),
                            albedo=P.to_device_f16(wide[0], dev), normal=P.to_device_f16(wide[1], dev),
                            arm=P.to_device_f16(wide[2], dev), position=P.to_device_f16(wide[3], dev))
    bench.WIDTH, bench.HEIGHT = bench.C5_WIDTH, bench.C5_ROWS
else:
    f = bench.frame_of(scene, synth.make_gbuffer(np.random.default_rng(bench.SEED + 1000), bench.WIDTH, bench.HEIGHT), bench.WIDTH, bench.HEIGHT)
    d = P.DeviceFrame.upload(f, dev)
color = torch.empty((bench.HEIGHT, bench.WIDTH, 4), dtype=torch.float16, device=dev)
thr = torch.empty_like(color)
verts, tris = synth.reference_light_volume()
dverts, dtris = torch.from_numpy(verts).to(dev), torch.from_numpy(tris.astype(np.int32)).to(dev)
forms = {"plain": d}
if not os.environ.get("VKR_C5"):
    masks, flags = P.light_coverage_with_flags(d, dverts, dtris)
    forms.update({"flags": dataclasses.replace(d, coverage=masks, coverage_flags=flags), "masks": dataclasses.replace(d, coverage=masks)})
only = os.environ.get("VKR_ONLY")
reps = int(os.environ.get("VKR_REPS", "10"))
res = {n: [] for n in forms}
for rnd in range(int(os.environ.get("VKR_ROUNDS", "4"))):  # interleaved: clocks settle over the first launches
    for name, fr in forms.items():
        if only and name != only:
            continue
        P.shade(fr, color, thr); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            P.shade(fr, color, thr)
        e1.record(); torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / reps, 3))
print(os.environ.get("VKR_HIP_LIB", "default"), f"{bench.WIDTH}x{bench.HEIGHT} lights {d.n_lights}", res,
      "ps per pixel and light:", round(min(res["plain"]) * 1e9 / (bench.WIDTH * bench.HEIGHT * max(d.n_lights, 1)), 3), flush=True)
