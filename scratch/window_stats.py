"""Per 8x8-pixel wave and light: do the cube-shadow texel indices of all lanes, for each of the
16 coordinate pairs, fit a small window around one lane's texel?"""
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
def waves(x): return x[:th*T, :tw*T].reshape(th, T, tw, T).transpose(0, 2, 1, 3).reshape(th, tw, T*T)
stats = {}
for li in range(0, L, 8):
    v = pos - lights["position"][li]
    C = [np.stack([v[..., k] - disk, v[..., k], v[..., k] + disk], -1) for k in range(3)]  # X,Y,Z candidates
    a = np.abs(v)
    amin = np.maximum(a - disk[..., None], 0); amax = a + disk[..., None]
    face = np.full(v.shape[:2], -1)
    for ax in (2, 1, 0):
        o = [k for k in range(3) if k != ax]
        ok = (amin[..., ax] >= amax[..., o[0]]) & (amin[..., ax] >= amax[..., o[1]]) & (face < 0)
        face[ok] = ax * 2 + (v[..., ax][ok] < 0)
    fw = waves(np.where(valid, face, -5))
    vw = waves(valid)
    # wave usable for the one-face path: all valid lanes agree on a face >= 0 (invalid lanes do not run the loop)
    fmax = np.where(vw, fw, -9).max(-1); fmin = np.where(vw, fw, 99).min(-1)
    uni = (fmax == fmin) & (fmax >= 0)
    nwaves = vw.any(-1).sum()
    fits = {}
    for ax in range(3):
        sel = uni & (fmax // 2 == ax)
        if not sel.any(): continue
        u_ax, w_ax = [(0, 1), (0, 2), (2, 1)][2 - ax] if False else ({2: (0, 1), 1: (0, 2), 0: (2, 1)}[ax])
        M = np.abs(C[ax])
        ok = {k: np.ones(sel.shape, bool) for k in ("c4", "m4", "m8x4", "c8")}
        for cand, axis in ((C[u_ax], "i"), (C[w_ax], "j")):
            for ai in range(3):
                for mi in range(3):
                    if ai == 1 and mi == 1: continue
                    t = np.clip(np.floor((cand[..., ai] / M[..., mi] * 0.5 + 0.5) * S), 0, S - 1).astype(np.int32)
                    tw_ = waves(t)
                    big = np.where(vw, tw_, -10**6).max(-1); small = np.where(vw, tw_, 10**6).min(-1)
                    anchor = tw_[..., 27]
                    ok["c4"] &= (small >= anchor - 1) & (big <= anchor + 2)
                    ok["c8"] &= (small >= anchor - 3) & (big <= anchor + 4)
                    ok["m4"] &= (big - small) <= 3
                    ok["m8x4"] &= (big - small) <= (7 if axis == "i" else 3)
        for k in ok:
            fits[k] = fits.get(k, 0) + (ok[k] & sel).sum()
    stats.setdefault("uniform", []).append(uni.sum() / nwaves)
    for k, n in fits.items():
        stats.setdefault(k, []).append(n / nwaves)
for k, v in stats.items():
    print(k, np.round(np.mean(v), 3), np.round(v, 2))
