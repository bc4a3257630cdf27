"""The frame rendered from the benchmark's mesh: how many (pixel, light) pairs could a range over the pixel's own tap
window settle, exactly and through cells of 4, 8 and 16 texels?  Pixel-level at tile centres; sizing only."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vulkanrenderer_amd import synth, passes as P
dev = torch.device("cuda:0")
W, H, L, RES = 3840, 2160, 64, 1024
f = synth.make_frame(bench.SEED, 64, 64, L, 8, 8)
lights = f.lights
verts, tris = synth.make_scene_mesh(128)
uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=0.25)
alb, nmap, arm = synth.make_material(np.random.default_rng(bench.SEED + 9), 64)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
gdraws = [dict(positions=t(verts, np.float32), triangles=t(tris, np.int32), texcoords=t(uv, np.float32), normals=t(nrm, np.float32),
               tangents=t(tan, np.float32), albedo=t(alb, np.uint8), normal=t(nmap, np.uint8), arm=t(arm, np.uint8), model=np.eye(4))]
scene = f.scene.copy()
scene["camera_projection"] = synth.perspective(60.0, W / H, 0.1, 100.0).astype(np.float32).ravel()
g = P.gbuffer(gdraws, scene, W, H, dev)
pos = g[3].cpu().numpy().view(np.float16).astype(np.float32)[4::8, 4::8, :3].reshape(-1, 3)
nr = g[1].cpu().numpy().view(np.float16).astype(np.float32)[4::8, 4::8, :3].reshape(-1, 3)
covered = g[0].cpu().numpy().view(np.float16)[4::8, 4::8, 3].reshape(-1) != 0
dl = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(dev)
cubes, _ = P.point_shadow_maps([(gdraws[0]["positions"], gdraws[0]["triangles"], np.eye(4))], dl, L, RES, tiled=False)
cubes = cubes.cpu().numpy() * 25.0
cam = np.asarray(scene["camera_position"], np.float32)
rng = np.random.default_rng(0)
sel = rng.choice(np.flatnonzero(covered), 2500, replace=False)
Pp = pos[sel]; N = nr[sel]
vd = np.linalg.norm(Pp - cam, axis=1); disk = (1 + vd / 25) / 25
names = ["exact window", "cells of 4", "cells of 8", "cells of 16", "blocks 2S at stride S (now, per pixel)"]
res = {k: [0, 0] for k in names}  # settled lit, settled shadow
tot = 0; truth = [0, 0, 0]
def mm(c, x0, y0, x1, y1):
    w = c[max(y0, 0):y1 + 1, max(x0, 0):x1 + 1]
    return w.min(), w.max()
for i in range(L):
    d = Pp - lights["position"][i]
    live = (N * -d).sum(1) > 0
    dist = np.linalg.norm(d, axis=1); cur = dist - 0.15
    a = np.abs(d); ax = a.argmax(1); M = a[np.arange(len(d)), ax]
    second = np.sort(a, axis=1)[:, 1]
    oneface = (M - disk) * (1 - 2**-20) >= second + disk
    sgn = d[np.arange(len(d)), ax] < 0; face = ax * 2 + sgn
    for p in np.flatnonzero(live & oneface):
        c = cubes[i, face[p]]
        dd = d[p]; m = dd[ax[p]]
        if ax[p] == 0: u, w_ = -dd[2] * np.sign(m), -dd[1]
        elif ax[p] == 1: u, w_ = dd[0], dd[2] * np.sign(m)
        else: u, w_ = dd[0] * np.sign(m), -dd[1]
        def win(x):
            q = [(x - disk[p]) / (M[p] - disk[p]), (x - disk[p]) / (M[p] + disk[p]), (x + disk[p]) / (M[p] - disk[p]), (x + disk[p]) / (M[p] + disk[p])]
            return int(np.clip(np.floor((min(q) * 0.5 + 0.5) * RES) - 1, 0, RES - 1)), int(np.clip(np.floor((max(q) * 0.5 + 0.5) * RES) + 1, 0, RES - 1))
        i0, i1 = win(u); j0, j1 = win(w_)
        tot += 1
        lo, hi = mm(c, i0, j0, i1, j1)
        def tally(name, lo, hi):
            if not (cur[p] > lo): res[name][0] += 1
            elif cur[p] > hi: res[name][1] += 1
        tally("exact window", lo, hi)
        truth[0 if not (cur[p] > lo) else (1 if cur[p] > hi else 2)] += 1
        for s in (4, 8, 16):
            tally(f"cells of {s}", *mm(c, i0 // s * s, j0 // s * s, i1 // s * s + s - 1, j1 // s * s + s - 1))
        width = max(i1 - i0, j1 - j0)
        k = int(width >> 4).bit_length(); S = 16 << k
        if k < 4: tally(names[4], *mm(c, (i0 // S) * S, (j0 // S) * S, (i0 // S) * S + 2 * S - 1, (j0 // S) * S + 2 * S - 1))
print("pairs (live, one face)", tot, "window lit / shadowed / mixed by the exact window:", [round(x / tot, 3) for x in truth])
for k, v in res.items(): print(f"{k:45s} lit {v[0] / tot:.3f} shadow {v[1] / tot:.3f} settled {(v[0] + v[1]) / tot:.3f}")
