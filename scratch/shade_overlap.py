"""Do two shading launches on two HIP streams overlap?  Two strips of VKR_HEIGHT rows, shaded one after the other on one
stream, then on two streams, against one launch of twice the rows."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vulkanrenderer_amd import passes as P, synth
dev = torch.device("cuda", 0)
H = int(os.environ.get("VKR_HEIGHT", "270"))
scene = bench.build_scene(dev, 64, bench.CUBE_RES, bench.DIR_RES)
f2 = bench.frame_of(scene, synth.make_gbuffer(np.random.default_rng(bench.SEED + 1000), bench.WIDTH, 2 * H), bench.WIDTH, 2 * H)
d2 = P.DeviceFrame.upload(f2, dev)
cut = lambda y0, y1: dataclasses.replace(d2, rows=y1 - y0, albedo=d2.albedo[y0:y1], normal=d2.normal[y0:y1], arm=d2.arm[y0:y1], position=d2.position[y0:y1])
a, b = cut(0, H), cut(H, 2 * H)
color = torch.empty((2 * H, bench.WIDTH, 4), dtype=torch.float16, device=dev)
thr = torch.empty_like(color)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)

def one_stream():
    P.shade(a, color[:H], thr[:H]); P.shade(b, color[H:], thr[H:])

def two_streams():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): P.shade(a, color[:H], thr[:H])
    with torch.cuda.stream(s2): P.shade(b, color[H:], thr[H:])
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)

for _ in range(2):
  print(f"rows {H}: one launch of {2 * H} rows {timed(lambda: P.shade(d2, color, thr))} ms, two launches on one stream {timed(one_stream)} ms, on two streams {timed(two_streams)} ms", flush=True)
