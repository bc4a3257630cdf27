"""Which (texel, region) pairs of the by-region irradiance kernel lose samples: counts from the workspace against a numpy classification."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import _lib, passes as P

env_n, n = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
env = (torch.rand((6, env_n, env_n, 4), device=dev) * 4.0).to(torch.float16)
out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=dev)
ws = P.irradiance_by_region(env, out)
torch.cuda.synchronize()
g = (env_n + 511) >> 9
n_regions, n_texels = 6 * g * g, 6 * n * n
padded = (n_texels + 63) // 64 * 64
part = ws[:n_regions * 4 * padded * 4].view(torch.float32).view(n_regions, 4, padded).cpu().numpy()
counts = part[:, 3, :n_texels]
tot = counts.sum(0)
bad = np.where(tot != 252 * 63)[0]
print("uncovered", P.irradiance_uncovered(ws), "bad texels", bad, tot[bad])

def face_dir(f, i, j):
    s, t = (2 * i + 1) / n - 1, (2 * j + 1) / n - 1
    return np.array([(1, -t, -s), (-1, -t, s), (s, 1, t), (s, -1, -t), (s, -t, 1), (-s, -t, -1)][f], dtype=np.float64)

phis, thetas = np.arange(252) * 0.025, np.arange(63) * 0.025
for tx in bad[:4]:
    f, rem = divmod(tx, n * n); j, i = divmod(rem, n)
    N = face_dir(f, i, j); N /= np.linalg.norm(N)
    right = np.array([N[2], 0, -N[0]]); right /= np.linalg.norm(right); up = np.cross(N, right)
    st, ct = np.sin(thetas)[None, :], np.cos(thetas)[None, :]
    d = (st * np.cos(phis)[:, None])[..., None] * right + (st * np.sin(phis)[:, None])[..., None] * up + (ct * np.ones((252, 1)))[..., None] * N
    ax = np.abs(d)
    z = (ax[..., 2] >= ax[..., 0]) & (ax[..., 2] >= ax[..., 1]); y = ~z & (ax[..., 1] >= ax[..., 0])
    face = np.where(z, np.where(d[..., 2] >= 0, 4, 5), np.where(y, np.where(d[..., 1] >= 0, 2, 3), np.where(d[..., 0] >= 0, 0, 1)))
    ma = np.where(z, ax[..., 2], np.where(y, ax[..., 1], ax[..., 0]))
    sc = np.select([face == 0, face == 1, face == 5], [-d[..., 2], d[..., 2], -d[..., 0]], d[..., 0])
    tc = np.select([face == 2, face == 3], [d[..., 2], -d[..., 2]], -d[..., 1])
    s, t = 0.5 * sc / ma + 0.5, 0.5 * tc / ma + 0.5
    i0 = np.clip(np.floor(s * env_n - 0.5).astype(int), 0, env_n - 1) >> 9; j0 = np.clip(np.floor(t * env_n - 0.5).astype(int), 0, env_n - 1) >> 9
    region = (j0 * g + i0) * 6 + face
    want = np.bincount(region.ravel(), minlength=n_regions)
    got = counts[:, tx].astype(int)
    diff = np.where(want != got)[0]
    print("texel", tx, (f, i, j), "N", N.round(4))
    for r in diff:
        hit = region == r
        aa, bb = np.where(hit.any(1))[0], np.where(hit.any(0))[0]
        print("   region", r, "face", r % 6, "cell", divmod(r // 6, g)[::-1], "want", want[r], "got", got[r], "phi idx", aa.min(), aa.max(), len(aa), "theta idx", bb.min(), bb.max())
