"""Shade the headline frame with and without the depth ranges: same bytes? how long each? how long the build?"""
import dataclasses, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import synth, passes as P, _lib

dev = torch.device("cuda:0")
W, H, L, RES = (int(v) for v in (sys.argv[1:5] + [3840, 2160, 64, 1024][len(sys.argv) - 1:]))
f = synth.make_frame(1, W, H, L, RES, 4096)
d = P.DeviceFrame.upload(f, dev)
plain = dataclasses.replace(d, cube_ranges=None)
outs = [[torch.empty((H, W, 4), dtype=torch.float16, device=dev) for _ in range(2)] for _ in range(2)]
def timed(fr, o, n=10):
    for _ in range(3): P.shade(fr, *o)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): P.shade(fr, *o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_plain = timed(plain, outs[0]); t_rng = timed(d, outs[1])
same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(outs[0], outs[1]))
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): P.cube_shadow_ranges(d.cube_shadow, d.cube_layout, d.cube_ranges)
e1.record(); torch.cuda.synchronize()
print(f"{W}x{H} {L} lights {RES}^2: plain {t_plain:.3f} ms, with ranges {t_rng:.3f} ms, same bytes {same}, build {e0.elapsed_time(e1) / 10:.3f} ms, "
      f"ranges {d.cube_ranges.numel() / 1e6:.1f} MB")
