"""How large is the cube-shadow texel footprint of a 16x16 (or 8x8) pixel tile, per light?"""
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
disk = (1 + np.linalg.norm(cam - pos, axis=-1) / 25) / 25
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res = {}
for li in range(0, L, 8):
    v = pos - lights["position"][li]
    a = np.abs(v)
    # major axis with margin: uniform face if min over +-disk of major >= max over +-disk of others
    amin = np.maximum(a - disk[..., None], 0); amax = a + disk[..., None]
    face = np.full(v.shape[:2], -1)
    for ax in (2, 1, 0):
        o = [k for k in range(3) if k != ax]
        ok = (amin[..., ax] >= amax[..., o[0]]) & (amin[..., ax] >= amax[..., o[1]]) & (face < 0)
        face[ok] = ax * 2 + (v[..., ax][ok] < 0)
    # texel coords of the centre and extent +-disk on the major face
    ax = np.argmax(a, -1)
    ma = np.take_along_axis(a, ax[..., None], -1)[..., 0]
    others = np.stack([np.where(ax == 0, v[..., 2], v[..., 0]), np.where(ax == 1, v[..., 2], v[..., 1])], -1)
    s = (others / ma[..., None] * 0.5 + 0.5) * S
    ext = (disk / np.maximum(ma - disk, 1e-6)) * 0.5 * S * (1 + np.abs(others).max(-1) / ma)  # conservative tap reach
    lo = s - ext[..., None]; hi = s + ext[..., None]
    th, tw = H // T, W // T
    def tiles(x, fn): return fn(fn(x[:th*T, :tw*T].reshape(th, T, tw, T), 3), 1)
    si = tiles(hi[..., 0], np.max) - tiles(lo[..., 0], np.min) + 1
    sj = tiles(hi[..., 1], np.max) - tiles(lo[..., 1], np.min) + 1
    span = np.maximum(si, sj)
    area = np.ceil(si / 4) * 4 * sj
    f_lo = tiles(np.where(valid, face, 99), np.min); f_hi = tiles(np.where(valid, face, -99), np.max)
    anyvalid = tiles(valid, np.max)
    uni = (f_lo == f_hi) & (f_lo >= 0) & anyvalid
    for lim in (32, 48, 64, 96, 128):
        res.setdefault(lim, []).append(((span <= lim) & uni).sum() / max(anyvalid.sum(), 1))
    for lim in (1024, 1536, 2048, 2496, 4096, 6144, 8192):
        res.setdefault("area%d" % lim, []).append(((area <= lim) & uni).sum() / max(anyvalid.sum(), 1))
    res.setdefault("mean_area_fit4096", []).append(area[(area <= 4096) & uni].mean())
    res.setdefault("uniform", []).append(uni.sum() / max(anyvalid.sum(), 1))
    res.setdefault("median_span", []).append(np.median(span[uni]))
for k, v in res.items():
    print(k, np.round(np.mean(v), 3))
