"""The frame rendered from the benchmark's mesh: would a plane plus residual bounds per block (instead of a plain min/max)
settle the pixels whose taps all see the slanted surface they lie on?  Blocks of 2S x 2S texels at stride S, S = 16 << level
chosen by the pixel's window as the shading kernel's old per-pixel test did; per tile: every live pixel settled.  Sizing only."""
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
tiles = lambda a: a.cpu().numpy().view(np.float16).astype(np.float32).reshape(H // 8, 8, W // 8, 8, 4).transpose(0, 2, 1, 3, 4).reshape(-1, 64, 4)
pos = tiles(g[3])[..., :3]; nr = tiles(g[1])[..., :3]; cov = tiles(g[0])[..., 3] != 0
dl = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(dev)
cubes, _ = P.point_shadow_maps([(gdraws[0]["positions"], gdraws[0]["triangles"], np.eye(4))], dl, L, RES, tiled=False)
cubes = cubes.cpu().numpy().astype(np.float64) * 25.0
cam = np.asarray(scene["camera_position"], np.float32)
rng = np.random.default_rng(0)
sel = rng.choice(np.flatnonzero(cov.all(1)), 600, replace=False)
cache = {}
def block_model(i, face, x0, y0, size):
    key = (i, face, x0, y0, size)
    if key not in cache:
        w = cubes[i, face][y0:y0 + size, x0:x0 + size]
        yy, xx = np.mgrid[0:w.shape[0], 0:w.shape[1]]
        xc, yc = xx - xx.mean(), yy - yy.mean()
        a = (w * xc).sum() / max((xc * xc).sum(), 1e-9); b = (w * yc).sum() / max((yc * yc).sum(), 1e-9)
        r = w - a * xx - b * yy
        cache[key] = (a, b, r.min(), r.max(), w.min(), w.max())
    return cache[key]
stat = {"pixel: min/max": 0, "pixel: plane": 0, "tile (every live pixel): min/max": 0, "tile (every live pixel): plane": 0, "tile (every live pixel): exact window": 0}
npx = 0; ntile = 0
for ti in sel:
    Pp = pos[ti]; N = nr[ti]
    vd = np.linalg.norm(Pp - cam, axis=1); disk = (1 + vd / 25) / 25
    for i in range(0, L, 4):
        d = Pp - lights["position"][i]
        live = (N * -d).sum(1) > 0
        if not live.any(): continue
        a_ = np.abs(d); ax = a_.argmax(1)
        if (ax != ax[0]).any(): continue
        A = ax[0]; M = a_[:, A]; m = d[:, A]
        if (np.sign(m) != np.sign(m[0])).any(): continue
        second = np.sort(a_, axis=1)[:, 1]
        if not ((M - disk) * (1 - 2**-20) >= second + disk).all(): continue
        face = A * 2 + (m[0] < 0); s = np.sign(m[0])
        if A == 0: u, w_ = -d[:, 2] * s, -d[:, 1]
        elif A == 1: u, w_ = d[:, 0], d[:, 2] * s
        else: u, w_ = d[:, 0] * s, -d[:, 1]
        def win(x):
            q = np.stack([(x - disk) / (M - disk), (x - disk) / (M + disk), (x + disk) / (M - disk), (x + disk) / (M + disk)])
            return (np.clip(np.floor((q.min(0) * 0.5 + 0.5) * RES) - 1, 0, RES - 1).astype(int), np.clip(np.floor((q.max(0) * 0.5 + 0.5) * RES) + 1, 0, RES - 1).astype(int))
        i0, i1 = win(u); j0, j1 = win(w_)
        cur = np.linalg.norm(d, axis=1) - 0.15
        ok = {"mm": True, "pl": True, "ex": True}
        for p in np.flatnonzero(live):
            width = max(i1[p] - i0[p], j1[p] - j0[p]); k = int(width >> 4).bit_length(); S = 16 << k
            npx += 1
            if k >= 4: ok = {q: False for q in ok}; continue
            x0, y0 = (i0[p] // S) * S, (j0[p] // S) * S
            a, b, rlo, rhi, lo, hi = block_model(i, face, x0, y0, 2 * S)
            xs = np.array([i0[p] - x0, i1[p] - x0]); ys = np.array([j0[p] - y0, j1[p] - y0])
            lb = rlo + (a * xs).min() + (b * ys).min(); ub = rhi + (a * xs).max() + (b * ys).max()
            s_mm = cur[p] > hi or not (cur[p] > lo)
            s_pl = cur[p] > min(ub, hi) or not (cur[p] > max(lb, lo))
            wdw = cubes[i, face][j0[p]:j1[p] + 1, i0[p]:i1[p] + 1]
            s_ex = cur[p] > wdw.max() or not (cur[p] > wdw.min())
            stat["pixel: min/max"] += s_mm; stat["pixel: plane"] += s_pl
            ok["mm"] &= s_mm; ok["pl"] &= s_pl; ok["ex"] &= s_ex
        ntile += 1
        stat["tile (every live pixel): min/max"] += ok["mm"]; stat["tile (every live pixel): plane"] += ok["pl"]; stat["tile (every live pixel): exact window"] += ok["ex"]
print("live pixels", npx, "tile-light pairs", ntile)
for k, v in stat.items(): print(f"{k:45s} {v / (npx if k.startswith('pixel') else ntile):.3f}")
