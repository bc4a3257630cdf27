"""Estimate, on the headline frame, how many (8x8 tile, light) pairs a block min/max test of the cube shadow map
would settle (all 20 taps shadowed, or none).  Approximate arithmetic; for sizing only."""
import sys
import numpy as np
sys.path.insert(0, ".")
from vulkanrenderer_amd import synth

W, H, L, RES = 3840, 2160, 64, 1024
rng = np.random.default_rng(1)
f = synth.make_frame(1, 64, 64, L, RES, 64)   # lights and cubes do not depend on the frame size... (rng order does)
lights, cubes = f.lights, f.cube_shadow * 25.0
cam = np.asarray(f.scene["camera_position"], np.float32)
# pixel positions at tile centres (8x8 tiles), as make_gbuffer lays them out
u = (np.arange(4, W, 8) + 0.5) / W; v = (np.arange(4, H, 8) + 0.5) / H
px = (-3 + 6 * u)[None, :].repeat(len(v), 0); pz = (-3 + 6 * v)[:, None].repeat(len(u), 1)
py = 0.3 * np.sin(1.7 * px) * np.cos(1.3 * pz)
P = np.stack([px, py, pz], -1).reshape(-1, 3)
vd = np.linalg.norm(P - cam, axis=1)
disk = (1 + vd / 25) / 25
for B in (16, 32, 64):
    nb = RES // B
    tot = concl = wide = 0
    for i in range(L):
        c = cubes[i].reshape(6, nb, B, nb, B)
        mn = c.min(axis=(2, 4)); mx = c.max(axis=(2, 4))
        d = P - lights["position"][i]
        dist = np.linalg.norm(d, axis=1); cur = dist - 0.15
        a = np.abs(d); ax = a.argmax(1); M = a[np.arange(len(d)), ax]
        ok = M > 3 * disk            # one-face path roughly: the major axis stays major
        sgn = d[np.arange(len(d)), ax] < 0
        face = ax * 2 + sgn
        o1 = np.where(ax == 0, 2, np.where(ax == 1, 0, 0)); o2 = np.where(ax == 0, 1, np.where(ax == 1, 2, 1))
        uu = d[np.arange(len(d)), o1]; ww = d[np.arange(len(d)), o2]
        def rng_(x):
            lo = np.minimum.reduce([(x - disk) / (M - disk), (x - disk) / (M + disk), (x + disk) / (M - disk), (x + disk) / (M + disk)])
            hi = np.maximum.reduce([(x - disk) / (M - disk), (x - disk) / (M + disk), (x + disk) / (M - disk), (x + disk) / (M + disk)])
            return np.clip(np.floor((lo * 0.5 + 0.5) * RES), 0, RES - 1).astype(int), np.clip(np.floor((hi * 0.5 + 0.5) * RES), 0, RES - 1).astype(int)
        i0, i1 = rng_(uu); j0, j1 = rng_(ww)   # orientation/sign of the face axes does not matter for a statistic
        bx0, bx1, by0, by1 = i0 // B, i1 // B, j0 // B, j1 // B
        narrow = (bx1 - bx0 <= 1) & (by1 - by0 <= 1)
        m_lo = np.minimum.reduce([mn[face, by0, bx0], mn[face, by0, bx1], mn[face, by1, bx0], mn[face, by1, bx1]])
        m_hi = np.maximum.reduce([mx[face, by0, bx0], mx[face, by0, bx1], mx[face, by1, bx0], mx[face, by1, bx1]])
        settled = narrow & ((cur > m_hi) | (cur <= m_lo))
        tot += ok.sum(); concl += (ok & settled).sum(); wide += (ok & ~narrow).sum()
    print(f"B={B}: one-face {tot / (L * len(P)):.3f}; settled {concl / tot:.3f}; window too wide {wide / tot:.3f}")

# overlapping blocks: level k has blocks of 2*S texels at stride S, S = 16 << k; a window no wider than S lies in one block
from scipy.ndimage import minimum_filter, maximum_filter
for S0 in (8, 16, 32):
    strides = [S0 << k for k in range(4)]
    tot = concl = 0; by_level = np.zeros(5)
    for i in range(L):
        lv = []
        for S in strides:
            c = cubes[i]
            nbk = RES // S
            # block b covers [b*S, b*S + 2S) clipped
            cm = c.reshape(6, nbk, S, nbk, S)
            mn1 = cm.min(axis=(2, 4)); mx1 = cm.max(axis=(2, 4))
            pad = lambda a, fill: np.pad(a, ((0, 0), (0, 1), (0, 1)), constant_values=fill)
            mnp, mxp = pad(mn1, np.inf), pad(mx1, -np.inf)
            mn2 = np.minimum.reduce([mnp[:, :-1, :-1], mnp[:, 1:, :-1], mnp[:, :-1, 1:], mnp[:, 1:, 1:]])
            mx2 = np.maximum.reduce([mxp[:, :-1, :-1], mxp[:, 1:, :-1], mxp[:, :-1, 1:], mxp[:, 1:, 1:]])
            lv.append((mn2, mx2))
        d = P - lights["position"][i]
        dist = np.linalg.norm(d, axis=1); cur = dist - 0.15
        a = np.abs(d); ax = a.argmax(1); M = a[np.arange(len(d)), ax]
        sgn = d[np.arange(len(d)), ax] < 0
        face = ax * 2 + sgn
        o1 = np.where(ax == 0, 2, 0); o2 = np.where(ax == 0, 1, np.where(ax == 1, 2, 1))
        uu = d[np.arange(len(d)), o1]; ww = d[np.arange(len(d)), o2]
        def rng_(x):
            q = [(x - disk) / (M - disk), (x - disk) / (M + disk), (x + disk) / (M - disk), (x + disk) / (M + disk)]
            lo = np.minimum.reduce(q); hi = np.maximum.reduce(q)
            return np.clip(np.floor((lo * 0.5 + 0.5) * RES), 0, RES - 1).astype(int), np.clip(np.floor((hi * 0.5 + 0.5) * RES), 0, RES - 1).astype(int)
        i0, i1 = rng_(uu); j0, j1 = rng_(ww)
        settled = np.zeros(len(d), bool); level = np.full(len(d), 4)
        for k, S in enumerate(strides):
            fits = (i1 // S - i0 // S <= 1) & (j1 // S - j0 // S <= 1) & (level == 4)   # lies in block (i0//S): [b*S, b*S+2S)
            level[fits] = k
            mn2, mx2 = lv[k]
            lo = mn2[face, j0 // S, i0 // S]; hi = mx2[face, j0 // S, i0 // S]
            settled |= fits & ((cur > hi) | (cur <= lo))
        tot += len(d); concl += settled.sum(); by_level += np.bincount(level, minlength=5)
    print(f"stride0={S0}: settled {concl / tot:.3f}; levels {np.round(by_level / tot, 3)}")
