"""Soak: 600 frames of the 4K / 64-light step with two frames in flight, every output compared with the first frame's
(the inputs do not change, so every frame must be byte-identical); then 200 frames through the renderer object
(layer 2) from draw calls on both frame slots."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from vulkanrenderer_amd import passes as P, synth
from vulkanrenderer_amd.strips import StripPlan, StripRenderer

dev = torch.device("cuda", 0)
scene = bench.build_scene(dev, 64, 1024, 4096)
f = bench.frame_of(scene, synth.make_gbuffer(np.random.default_rng(bench.SEED + 1000), 3840, 2160), 3840, 2160)
d = P.DeviceFrame.upload(f, dev)
r = StripRenderer(StripPlan.make(0, 1, 2160), 3840, device=dev, frames_in_flight=2)
out0 = r.render(d); r.gather(0); r.synchronize(); torch.cuda.synchronize()
first = out0.clone()
bad = 0
t0 = time.time()
for k in range(600):
    out = r.render(d)
    r.gather(0)
    if k % 10 == 9:
        r.synchronize(); torch.cuda.synchronize()
        bad += int(not torch.equal(out, first))
torch.cuda.synchronize()
print(f"600 frames in {time.time() - t0:.2f} s, {bad} of 60 checked frames differ", flush=True)
sys.exit(1 if bad else 0)
