"""Tile-level classification estimate: per (8x8 tile, light), from the tile's position bounding box, can one range prove
that every pixel's 20 taps are all in shadow / all lit?  Sizing only (approximate arithmetic)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import synth
W, H, L, RES = 3840, 2160, 64, 1024
f = synth.make_frame(1, W, H, L, RES, 64)
lights, cubes = f.lights, f.cube_shadow * 25.0
cam = np.asarray(f.scene["camera_position"], np.float32)
pos = f.position.view(np.float16).astype(np.float32)[..., :3].reshape(H // 8, 8, W // 8, 8, 3)
nrm = f.normal.view(np.float16).astype(np.float32)[..., :3].reshape(H // 8, 8, W // 8, 8, 3)
shaded = (nrm != 0).any(-1)                       # [ty, 8, tx, 8]
big = np.where(shaded[..., None], pos, np.inf); small = np.where(shaded[..., None], pos, -np.inf)
pmin = big.min(axis=(1, 3)).reshape(-1, 3); pmax = small.max(axis=(1, 3)).reshape(-1, 3)
any_shaded = shaded.any(axis=(1, 3)).reshape(-1)
rng = np.random.default_rng(0)
sel = rng.choice(np.flatnonzero(any_shaded), 20000, replace=False)
pmin, pmax = pmin[sel], pmax[sel]
vdmax = np.linalg.norm(np.maximum(np.abs(pmin - cam), np.abs(pmax - cam)), axis=1); disk = (1 + vdmax / 25) / 25
# block ranges: level k, stride S=16<<k, block 2S
lv = []
for k in range(4):
    S = 16 << k; nb = RES // S
    cm = cubes.reshape(L, 6, nb, S, nb, S)
    mn1 = cm.min(axis=(3, 5)); mx1 = cm.max(axis=(3, 5))
    pad = lambda a, fill: np.pad(a, ((0, 0), (0, 0), (0, 1), (0, 1)), constant_values=fill)
    a, b = pad(mn1, np.inf), pad(mx1, -np.inf)
    lv.append((np.minimum.reduce([a[:, :, :-1, :-1], a[:, :, 1:, :-1], a[:, :, :-1, 1:], a[:, :, 1:, 1:]]),
               np.maximum.reduce([b[:, :, :-1, :-1], b[:, :, 1:, :-1], b[:, :, :-1, 1:], b[:, :, 1:, 1:]])))
lv8 = []
for k in range(5):
    S = 8 << k; nb = RES // S
    cm = cubes.reshape(L, 6, nb, S, nb, S)
    mn1 = cm.min(axis=(3, 5)); mx1 = cm.max(axis=(3, 5))
    pad = lambda a, fill: np.pad(a, ((0, 0), (0, 0), (0, 1), (0, 1)), constant_values=fill)
    a, b = pad(mn1, np.inf), pad(mx1, -np.inf)
    lv8.append((np.minimum.reduce([a[:, :, :-1, :-1], a[:, :, 1:, :-1], a[:, :, :-1, 1:], a[:, :, 1:, 1:]]),
                np.maximum.reduce([b[:, :, :-1, :-1], b[:, :, 1:, :-1], b[:, :, :-1, 1:], b[:, :, 1:, 1:]])))
skip4 = lit4 = 0
skip5 = lit5 = 0; trips = []
cell_mn = cubes.reshape(L, 6, RES // 16, 16, RES // 16, 16).min(axis=(3, 5)); cell_mx = cubes.reshape(L, 6, RES // 16, 16, RES // 16, 16).max(axis=(3, 5))
lv2 = []   # stride s = 8 << k, blocks of 4s: a window narrower than 3s lies in the block of its first column and row
for k in range(4):
    st = 8 << k; nb = RES // st
    cm = cubes.reshape(L, 6, nb, st, nb, st)
    mn1 = cm.min(axis=(3, 5)); mx1 = cm.max(axis=(3, 5))
    a = np.pad(mn1, ((0, 0), (0, 0), (0, 3), (0, 3)), constant_values=np.inf); b = np.pad(mx1, ((0, 0), (0, 0), (0, 3), (0, 3)), constant_values=-np.inf)
    lv2.append((np.minimum.reduce([a[:, :, dy:dy + nb, dx:dx + nb] for dy in range(4) for dx in range(4)]),
                np.maximum.reduce([b[:, :, dy:dy + nb, dx:dx + nb] for dy in range(4) for dx in range(4)])))
n = len(sel); tot = n * L; skip = lit = 0; skip2 = lit2 = 0; skip3 = lit3 = 0
for i in range(L):
    lo3, hi3 = pmin - lights["position"][i], pmax - lights["position"][i]
    amin = np.where((lo3 <= 0) & (hi3 >= 0), 0, np.minimum(np.abs(lo3), np.abs(hi3))); amax = np.maximum(np.abs(lo3), np.abs(hi3))
    cur_min = np.sqrt((amin ** 2).sum(1)) - 0.15; cur_max = np.sqrt((amax ** 2).sum(1)) - 0.15
    ax = amin.argmax(1); idx = np.arange(n)
    Mmin = amin[idx, ax] - disk; Mmax = amax[idx, ax] + disk
    others = np.stack([amax[idx, (ax + 1) % 3], amax[idx, (ax + 2) % 3]], 1).max(1) + disk
    one = (Mmin * (1 - 2**-20) >= others) & (Mmin > 0)
    neg = hi3[idx, ax] < 0; face = ax * 2 + neg
    o1 = np.where(ax == 0, 2, 0); o2 = np.where(ax == 0, 1, np.where(ax == 1, 2, 1))
    def win(o):
        ulo, uhi = lo3[idx, o] - disk, hi3[idx, o] + disk
        q = [ulo / Mmin, ulo / Mmax, uhi / Mmin, uhi / Mmax]
        a = np.floor((np.minimum.reduce(q) * 0.5 + 0.5) * RES); b = np.floor((np.maximum.reduce(q) * 0.5 + 0.5) * RES)
        return np.clip(a, 0, RES - 1).astype(int), np.clip(b, 0, RES - 1).astype(int)
    with np.errstate(all="ignore"):
        i0, i1 = win(o1); j0, j1 = win(o2)
    width = np.maximum(i1 - i0, j1 - j0)
    level = np.array([int(w >> 4).bit_length() for w in width])
    ok = one & (level < 4)
    s_skip = np.zeros(n, bool); s_lit = np.zeros(n, bool)
    for k in range(4):
        m = ok & (level == k); S = 16 << k
        lo = lv[k][0][i, face[m], j0[m] // S, i0[m] // S]; hi = lv[k][1][i, face[m], j0[m] // S, i0[m] // S]
        s_skip[m] = cur_min[m] > hi; s_lit[m] = cur_max[m] <= lo
    skip += s_skip.sum(); lit += s_lit.sum()
    level3 = np.full(n, 9)   # smallest level whose block at the window's first column and row holds the window, by alignment
    for k in range(3, -1, -1):
        S = 16 << k
        fits = ((i0 // S) * S + 2 * S > i1) & ((j0 // S) * S + 2 * S > j1)
        level3[fits] = k
    ok3 = one & (level3 < 4)
    u_skip = np.zeros(n, bool); u_lit = np.zeros(n, bool)
    for k in range(4):
        m = ok3 & (level3 == k); S = 16 << k
        lo = lv[k][0][i, face[m], j0[m] // S, i0[m] // S]; hi = lv[k][1][i, face[m], j0[m] // S, i0[m] // S]
        u_skip[m] = cur_min[m] > hi; u_lit[m] = cur_max[m] <= lo
    skip3 += u_skip.sum(); lit3 += u_lit.sum()
    # second chance for what the blocks leave undecided: min/max over the 16x16 cells the window touches (at most 6x6 of them)
    w_skip = u_skip.copy(); w_lit = u_lit.copy()
    for t in np.flatnonzero(one & ~(u_skip | u_lit)):
        cx0, cx1, cy0, cy1 = i0[t] // 16, i1[t] // 16, j0[t] // 16, j1[t] // 16
        if cx1 - cx0 > 5 or cy1 - cy0 > 5: continue
        lo = cell_mn[i, face[t], cy0:cy1 + 1, cx0:cx1 + 1].min(); hi = cell_mx[i, face[t], cy0:cy1 + 1, cx0:cx1 + 1].max()
        w_skip[t] = cur_min[t] > hi; w_lit[t] = cur_max[t] <= lo
        trips.append((cx1 - cx0 + 1) * (cy1 - cy0 + 1))
    skip5 += w_skip.sum(); lit5 += w_lit.sum()
    level4 = np.full(n, 9)
    for k in range(4, -1, -1):
        S = 8 << k
        fits = ((i0 // S) * S + 2 * S > i1) & ((j0 // S) * S + 2 * S > j1)
        level4[fits] = k
    ok4 = one & (level4 < 5)
    v_skip = np.zeros(n, bool); v_lit = np.zeros(n, bool)
    for k in range(5):
        m = ok4 & (level4 == k); S = 8 << k
        lo = lv8[k][0][i, face[m], j0[m] // S, i0[m] // S]; hi = lv8[k][1][i, face[m], j0[m] // S, i0[m] // S]
        v_skip[m] = cur_min[m] > hi; v_lit[m] = cur_max[m] <= lo
    skip4 += v_skip.sum(); lit4 += v_lit.sum()
    level2 = np.full(n, 9)
    for k in range(3, -1, -1):
        level2[width < 3 * (8 << k)] = k
    ok2 = one & (level2 < 4)
    t_skip = np.zeros(n, bool); t_lit = np.zeros(n, bool)
    for k in range(4):
        m = ok2 & (level2 == k); st = 8 << k
        lo = lv2[k][0][i, face[m], j0[m] // st, i0[m] // st]; hi = lv2[k][1][i, face[m], j0[m] // st, i0[m] // st]
        t_skip[m] = cur_min[m] > hi; t_lit[m] = cur_max[m] <= lo
    skip2 += t_skip.sum(); lit2 += t_lit.sum()
print(f"by alignment + cells of the window: skip {skip5 / tot:.3f}  lit {lit5 / tot:.3f}  general {(tot - skip5 - lit5) / tot:.3f}; cells per second chance: mean {np.mean(trips):.1f}, 90th percentile {np.percentile(trips, 90):.0f}")
print(f"by alignment, five levels from 8: skip {skip4 / tot:.3f}  lit {lit4 / tot:.3f}  general {(tot - skip4 - lit4) / tot:.3f}")
print(f"by alignment: skip {skip3 / tot:.3f}  lit {lit3 / tot:.3f}  general {(tot - skip3 - lit3) / tot:.3f}")
print(f"denser blocks: skip {skip2 / tot:.3f}  lit {lit2 / tot:.3f}  general {(tot - skip2 - lit2) / tot:.3f}")
print(f"tile-level: skip {skip / tot:.3f}  lit {lit / tot:.3f}  general {(tot - skip - lit) / tot:.3f}")
