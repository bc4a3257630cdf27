"""What happens to the (wave, light) pairs of the headline frame (library built with -DVKR_EXPERIMENT_COUNT)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanrenderer_amd import synth, passes as P, _lib
dev = torch.device("cuda:0")
W, H, L, RES = 3840, 2160, 64, 1024
f = synth.make_frame(1, W, H, L, RES, 4096)
d = P.DeviceFrame.upload(f, dev)
o = [torch.empty((H, W, 4), dtype=torch.float16, device=dev) for _ in range(2)]
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
lib.vkr_experiment_shade_counts(buf, 1)
P.shade(d, *o); torch.cuda.synchronize()
lib.vkr_experiment_shade_counts(buf, 1)
c = list(buf)
names = ["no live lane", "one face, settled by a range", "one face, taps read", "edge or corner, taps read", "all live lanes fully shadowed (no BRDF)", "BRDF evaluated"]
tot = c[0] + c[1] + c[2] + c[3]
for n, v in zip(names, c): print(f"{n:45s} {v:10d} {v / tot:.3f}")
