"""How many (tile, light) pairs could any range structure settle?  Exact min/max over the tap window itself (the ceiling),
against block structures of several shapes.  Lane-level at tile centres; sizing only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import synth
W, H, L, RES = 3840, 2160, 64, 1024
f = synth.make_frame(1, W, H, L, RES, 64)
lights, cubes = f.lights, f.cube_shadow * 25.0
cam = np.asarray(f.scene["camera_position"], np.float32)
pos = f.position.view(np.float16).astype(np.float32)[4::8, 4::8, :3].reshape(-1, 3)
nrm = f.normal.view(np.float16).astype(np.float32)[4::8, 4::8, :3].reshape(-1, 3)
keep = (nrm != 0).any(1)
rng = np.random.default_rng(0)
sel = rng.choice(np.flatnonzero(keep), 3000, replace=False)
P = pos[sel]; N = nrm[sel]
vd = np.linalg.norm(P - cam, axis=1); disk = (1 + vd / 25) / 25
res = {k: 0 for k in ("ideal", "centre_bound_S_2S", "exact_S_2S", "exact_halfS_2S", "exact_2x2_finer", "centre_2x2_finer")}
tot = 0
def blockmm(c, x0, y0, size):
    w = c[max(y0, 0):y0 + size, max(x0, 0):x0 + size]
    return w.min(), w.max()
for i in range(L):
    d = P - lights["position"][i]
    live = (N * -d).sum(1) > 0
    dist = np.linalg.norm(d, axis=1); cur = dist - 0.15
    a = np.abs(d); ax = a.argmax(1); M = a[np.arange(len(d)), ax]
    second = np.sort(a, axis=1)[:, 1]
    oneface = (M - disk) * (1 - 2**-20) >= second + disk
    sgn = d[np.arange(len(d)), ax] < 0; face = ax * 2 + sgn
    o1 = np.where(ax == 0, 2, 0); o2 = np.where(ax == 0, 1, np.where(ax == 1, 2, 1))
    uu = d[np.arange(len(d)), o1]; ww = d[np.arange(len(d)), o2]
    for p in np.flatnonzero(live & oneface):
        c = cubes[i, face[p]]
        def win(x):
            q = [(x - disk[p]) / (M[p] - disk[p]), (x - disk[p]) / (M[p] + disk[p]), (x + disk[p]) / (M[p] - disk[p]), (x + disk[p]) / (M[p] + disk[p])]
            return int(np.clip(np.floor((min(q) * 0.5 + 0.5) * RES), 0, RES - 1)), int(np.clip(np.floor((max(q) * 0.5 + 0.5) * RES), 0, RES - 1))
        i0, i1 = win(uu[p]); j0, j1 = win(ww[p])
        tot += 1
        def settled(lo, hi): return cur[p] > hi or not (cur[p] > lo)
        w = c[j0:j1 + 1, i0:i1 + 1]
        res["ideal"] += settled(w.min(), w.max())
        width = max(i1 - i0, j1 - j0)
        # exact window, stride S block 2S
        k = max(int(width >> 4).bit_length(), 0); S = 16 << k
        if k < 4: res["exact_S_2S"] += settled(*blockmm(c, (i0 // S) * S, (j0 // S) * S, 2 * S))
        # centre bound (+3 slack both sides)
        q0u, q0w = uu[p] / M[p], ww[p] / M[p]
        K = 512 * disk[p] / (M[p] - disk[p])
        hw = int(max(abs(q0u) * K + K + 3, abs(q0w) * K + K + 3))
        ci, cj = int((q0u * 0.5 + 0.5) * RES), int((q0w * 0.5 + 0.5) * RES)
        kk = int(hw >> 3).bit_length(); S2 = 16 << kk
        il, jl = max(ci - hw, 0), max(cj - hw, 0)
        if kk < 4: res["centre_bound_S_2S"] += settled(*blockmm(c, (il // S2) * S2, (jl // S2) * S2, 2 * S2))
        # stride S/2, block 2S: window < 1.5 S fits: choose smallest S = 8 << k with width < 1.5 S
        for k3 in range(5):
            S3 = 8 << k3
            if width < 1.5 * S3:
                h = S3 // 2
                res["exact_halfS_2S"] += settled(*blockmm(c, (i0 // h) * h, (j0 // h) * h, 2 * S3)); break
        # 2x2 blocks of the finer level: blocks of size S at stride S/2 (level k-1), window < S
        if k < 4:
            h = S // 2
            x0, y0 = (i0 // h) * h, (j0 // h) * h
            res["exact_2x2_finer"] += settled(*blockmm(c, x0, y0, h + S))
        if kk < 4:
            h = S2 // 2
            x0, y0 = (il // h) * h, (jl // h) * h
            res["centre_2x2_finer"] += settled(*blockmm(c, x0, y0, h + S2))
        for s0 in (4, 8):
            for k4 in range(5 if s0 == 4 else 4):
                st = s0 << k4
                if 2 * hw < 3 * st:
                    res.setdefault(f"centre_stride{s0}_block{4 * s0}", 0)
                    res[f"centre_stride{s0}_block{4 * s0}"] += settled(*blockmm(c, (il // st) * st, (jl // st) * st, 4 * st)); break
        hw2 = max(hw - 2, 0)   # slack of 1 instead of 3 (not proven: to see what the slack costs)
        il2, jl2 = max(ci - hw2, 0), max(cj - hw2, 0)
        for k4 in range(4):
            st = 8 << k4
            if 2 * hw2 < 3 * st:
                res.setdefault("centre_slack1_stride8_block32", 0)
                res["centre_slack1_stride8_block32"] += settled(*blockmm(c, (il2 // st) * st, (jl2 // st) * st, 4 * st)); break
print("one-face live pairs", tot)
for k, v in res.items(): print(f"{k:22s} {v / tot:.3f}")
