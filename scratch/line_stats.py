"""Distinct 128-byte lines a wave's 64 lanes touch per cube-shadow tap, for row-major and tiled
depth-map layouts (bench scene, subsampled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanrenderer_amd import synth
import bench
W, H, L, S = 3840, 2160, 64, 1024
rng = np.random.default_rng(bench.SEED)
lights = synth.make_lights(rng, L)
albedo, normal, arm, position = synth.make_gbuffer(np.random.default_rng(bench.SEED + 1000), W, H)
pos = position.view(np.float16).astype(np.float32)[..., :3]
valid = (normal.view(np.float16).astype(np.float32)[..., :3] ** 2).sum(-1) > 0
cam = np.array([2, 1, -4], np.float32)
disk = ((1 + np.linalg.norm(cam - pos, axis=-1) / 25) / 25).astype(np.float32)
T = 8
th, tw = H // T, W // T
def waves(x): return x[:th*T, :tw*T].reshape(th, T, tw, T).transpose(0, 2, 1, 3).reshape(th, tw, T*T)[::3, ::3]
GRID = [(2,2,2),(2,0,2),(0,0,2),(0,2,2),(2,2,0),(2,0,0),(0,0,0),(0,2,0),(2,2,1),(2,0,1),(0,0,1),(0,2,1),(2,1,2),(0,1,2),(2,1,0),(0,1,0),(1,2,2),(1,0,2),(1,0,0),(1,2,0)]
layouts = {"row-major": lambda i, j: (j * S + i) >> 5,
           "8x4": lambda i, j: (j >> 2) * (S >> 3) + (i >> 3),
           "4x8": lambda i, j: (j >> 3) * (S >> 2) + (i >> 2),
           "16x2": lambda i, j: (j >> 1) * (S >> 4) + (i >> 4)}
tot = {k: 0 for k in layouts}; n = 0
vw = waves(valid)
for li in range(0, L, 8):
    v = pos - lights["position"][li]
    C = [np.stack([v[..., k] - disk, v[..., k], v[..., k] + disk], -1) for k in range(3)]
    for g in GRID[::3]:
        d = np.stack([C[0][..., g[0]], C[1][..., g[1]], C[2][..., g[2]]], -1)
        a = np.abs(d); ax = np.argmax(a[..., ::-1], -1); ax = 2 - ax  # ties z over y over x
        ma = np.take_along_axis(a, ax[..., None], -1)[..., 0]
        neg = np.take_along_axis(d, ax[..., None], -1)[..., 0] < 0
        x, y, z = d[..., 0], d[..., 1], d[..., 2]
        sc = np.where(ax == 2, np.where(neg, -x, x), np.where(ax == 1, x, np.where(neg, z, -z)))
        tc = np.where(ax == 1, np.where(neg, -z, z), -y)
        face = ax * 2 + neg
        face = np.where(ax == 2, 4 + neg, np.where(ax == 1, 2 + neg, neg))
        with np.errstate(all="ignore"):
            i = np.clip(np.floor((sc / ma * 0.5 + 0.5) * S), 0, S - 1).astype(np.int64)
            j = np.clip(np.floor((tc / ma * 0.5 + 0.5) * S), 0, S - 1).astype(np.int64)
        for name, fn in layouts.items():
            line = waves(face * (S * S // 32) + fn(i, j))
            line = np.where(vw, line, -1)
            srt = np.sort(line, -1)
            distinct = (np.diff(srt, axis=-1) != 0).sum(-1) + 1 - (srt[..., 0] == -1)  # drop the invalid marker
            tot[name] += distinct[vw.any(-1)].sum()
        n += vw.any(-1).sum()
for k, v in tot.items():
    print(f"{k:10s} {v / n:6.2f} lines per wave-tap")
