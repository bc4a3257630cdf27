"""Profiling helper: builds the benchmark frame and runs only the shade kernel a few times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vulkanrenderer_amd import passes as P
dev = torch.device("cuda", 0)
L = int(os.environ.get("VKR_LIGHTS", "64"))
f, d = bench.build_inputs(0, dev, bench.HEIGHT, L, bench.CUBE_RES, bench.DIR_RES)
color = torch.empty((bench.HEIGHT, bench.WIDTH, 4), dtype=torch.float16, device=dev)
thr = torch.empty_like(color)
for _ in range(int(os.environ.get("VKR_REPS", "3"))):
    P.shade(d, color, thr)
torch.cuda.synchronize()
print("done")
