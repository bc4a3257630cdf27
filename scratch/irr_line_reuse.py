"""How much cache-line reuse is there BETWEEN neighbouring output texels of the irradiance pass?

For a block of B x B adjacent output texels of one face (what one XCD could be given to work on at the same
time) and one chunk of 16 phi slices (what the workgroups of the kernel have in flight together), count the
128-byte lines (4x4-texel tiles of the 2048^2 RGBA16F environment cube) that the bilinear lookups of all
texels of the block touch, against the lines each texel touches on its own.  ratio = sum over texels of
own distinct lines / distinct lines of the block: what a cache shared by the block could save at best."""
import numpy as np

ENV, OUT = 2048, 64
PI = np.float32(3.14159265359)

def face_dir(f, i, j, n):
    s = 2.0 * (i + 0.5) / n - 1.0
    t = 2.0 * (j + 0.5) / n - 1.0
    one = np.ones_like(s)
    return [(one, -t, -s), (-one, -t, s), (s, one, t), (s, -one, -t), (s, -t, one), (-s, -t, -one)][f]

def lines_of(d):
    """Line ids (face, tile y, tile x) of the 2x2 footprints of directions d [..., 3] (edge taps clamped to the face)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    face = np.where((az >= ax) & (az >= ay), np.where(z >= 0, 4, 5), np.where(ay >= ax, np.where(y >= 0, 2, 3), np.where(x >= 0, 0, 1)))
    ma = np.maximum(np.maximum(ax, ay), az)
    sc = np.choose(face, [-z, z, x, x, x, -x]); tc = np.choose(face, [-y, -y, z, -z, -y, -y])
    u = 0.5 * (sc / ma + 1.0) * ENV - 0.5; v = 0.5 * (tc / ma + 1.0) * ENV - 0.5
    i0, j0 = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
    ids = []
    for di in (0, 1):
        for dj in (0, 1):
            ii, jj = np.clip(i0 + di, 0, ENV - 1), np.clip(j0 + dj, 0, ENV - 1)
            ids.append((face * (ENV // 4) + jj // 4) * (ENV // 4) + ii // 4)
    return np.stack(ids, -1)

phis = np.arange(252, dtype=np.float32) * np.float32(0.025)
thetas = np.arange(63, dtype=np.float32) * np.float32(0.025)

def sample_dirs(N, phi_idx):
    N = N / np.linalg.norm(N)
    right = np.cross([0.0, 1.0, 0.0], N); right /= np.linalg.norm(right)
    up = np.cross(N, right); up /= np.linalg.norm(up)
    ph, th = np.meshgrid(phis[phi_idx], thetas, indexing="ij")
    tx, ty, tz = np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)
    return tx[..., None] * right + ty[..., None] * up + tz[..., None] * N

for B in (4, 8, 16):
    for (f, i0, j0) in ((4, 24, 24), (0, 0, 0), (2, 40, 8)):
        for chunk, label in ((range(0, 16), "16 phi slices"), (range(0, 252), "all 252 slices")):
            own, block = 0, set()
            for j in range(j0, j0 + B):
                for i in range(i0, i0 + B):
                    d = np.array([c[()] if np.ndim(c) == 0 else c for c in face_dir(f, np.float64(i), np.float64(j), OUT)], dtype=np.float64)
                    ids = np.unique(lines_of(sample_dirs(d, list(chunk))))
                    own += len(ids)
                    block.update(ids.tolist())
            print(f"block {B}x{B} at face {f} ({i0},{j0}), {label}: lines per texel {own / B / B:.0f}, block distinct {len(block)}, reuse ratio {own / len(block):.2f}")

print("\nworking set and reuse for one XCD's block, by how many phi slices the block's workgroups have in flight together")
for (bw, bh) in ((16, 16), (16, 12), (8, 8), (32, 8)):
    for nsl in (1, 2, 4, 8, 16):
        f, i0, j0 = 4, 16, 16
        own, block = 0, set()
        for j in range(j0, j0 + bh):
            for i in range(i0, i0 + bw):
                d = np.array(face_dir(f, np.float64(i), np.float64(j), OUT), dtype=np.float64)
                ids = np.unique(lines_of(sample_dirs(d, list(range(32, 32 + nsl)))))
                own += len(ids)
                block.update(ids.tolist())
        print(f"block {bw}x{bh}, {nsl} slices: distinct lines {len(block)} = {len(block) * 128 / 2**20:.1f} MiB, reuse ratio {own / len(block):.2f}")
