"""Bounding window (texels) of all 20 taps of all lanes of an 8x8 wave on its common face."""
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
vw = waves(valid)
allw = []; allh = []; nuni = 0; ntot = 0
for li in range(0, L, 8):
    v = pos - lights["position"][li]
    C = [np.stack([v[..., k] - disk, v[..., k], v[..., k] + disk], -1) for k in range(3)]
    a = np.abs(v)
    amin = np.maximum(a - disk[..., None], 0); amax = a + disk[..., None]
    face = np.full(v.shape[:2], -1)
    for ax in (2, 1, 0):
        o = [k for k in range(3) if k != ax]
        ok = (amin[..., ax] >= amax[..., o[0]]) & (amin[..., ax] >= amax[..., o[1]]) & (face < 0)
        face[ok] = ax * 2 + (v[..., ax][ok] < 0)
    fw = waves(face)
    fmax = np.where(vw, fw, -9).max(-1); fmin = np.where(vw, fw, 99).min(-1)
    uni = (fmax == fmin) & (fmax >= 0)
    ntot += vw.any(-1).sum(); nuni += uni.sum()
    for ax in range(3):
        sel = uni & (fmax // 2 == ax)
        if not sel.any(): continue
        u_ax, w_ax = {2: (0, 1), 1: (0, 2), 0: (2, 1)}[ax]
        M = np.abs(C[ax])
        ext = []
        for cand in (C[u_ax], C[w_ax]):
            lo = None; hi = None
            for ai in (0, 2):
                for mi in (0, 2):
                    with np.errstate(all="ignore"):
                        t = np.clip(np.floor((cand[..., ai] / M[..., mi] * 0.5 + 0.5) * S), 0, S - 1).astype(np.int32)
                    t = waves(t)
                    b = np.where(vw, t, -10**6).max(-1); s_ = np.where(vw, t, 10**6).min(-1)
                    lo = s_ if lo is None else np.minimum(lo, s_); hi = b if hi is None else np.maximum(hi, b)
            ext.append((lo[sel], hi[sel]))
        wt = ((ext[0][1] - (ext[0][0] & ~3)) & ~3) + 4
        ht = ext[1][1] - ext[1][0] + 1
        allw.append(wt); allh.append(ht)
wt = np.concatenate(allw); ht = np.concatenate(allh)
area = wt * ht
print("uniform-face wave-lights:", nuni / ntot)
print("window width  mean %.1f median %d p90 %d" % (wt.mean(), np.median(wt), np.percentile(wt, 90)))
print("window height mean %.1f median %d p90 %d" % (ht.mean(), np.median(ht), np.percentile(ht, 90)))
print("window bytes  mean %.0f median %.0f p90 %.0f p99 %.0f" % (area.mean() * 4, np.median(area) * 4, np.percentile(area, 90) * 4, np.percentile(area, 99) * 4))
for cap in (512, 1024, 1536, 2496):
    print("fits %d texels: %.3f" % (cap, (area <= cap).mean()))
# number of 64-lane dwordx4 instructions to fetch it (whole rows per instruction, 64 / (wt/4) rows each)
qw = wt // 4
rows_per = np.maximum(64 // qw, 1)
n_instr = np.ceil(ht / rows_per)
print("dwordx4 wave-loads per window: mean %.2f median %d p90 %d" % (n_instr.mean(), np.median(n_instr), np.percentile(n_instr, 90)))
