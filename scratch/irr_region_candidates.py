"""How many (phi, theta) cells the by-region irradiance kernel has to look at per sample that lands in the region, for
several ways of bounding the region in a texel's tangent frame (numpy model; 2048^2 faces, 64^2 cube, regions of 512^2)."""
import numpy as np
rng = np.random.default_rng(1)
ENV, N, RS = 2048, 64, 512
nphi, ntheta, step = 252, 63, 0.025
phis, thetas = np.arange(nphi) * step, np.arange(ntheta) * step

def face_point(f, s, t):
    o = np.ones_like(s)
    return np.stack([(o, -t, -s), (-o, -t, s), (s, o, t), (s, -o, -t), (s, -t, o), (-s, -t, -o)][f], -1)

def cube_face_st(d):
    ax = np.abs(d)
    z = (ax[..., 2] >= ax[..., 0]) & (ax[..., 2] >= ax[..., 1]); y = ~z & (ax[..., 1] >= ax[..., 0]); x = ~z & ~y
    face = np.where(z, np.where(d[..., 2] >= 0, 4, 5), np.where(y, np.where(d[..., 1] >= 0, 2, 3), np.where(d[..., 0] >= 0, 0, 1)))
    ma = np.where(z, ax[..., 2], np.where(y, ax[..., 1], ax[..., 0]))
    sc = np.select([face == 0, face == 1, face == 5], [-d[..., 2], d[..., 2], -d[..., 0]], d[..., 0])
    tc = np.select([face == 2, face == 3], [d[..., 2], -d[..., 2]], -d[..., 1])
    return face, 0.5 * sc / ma + 0.5, 0.5 * tc / ma + 0.5

def arc_of(hit):
    as_ = np.where(hit.any(1))[0]
    gaps = np.diff(np.concatenate([as_, [as_[0] + nphi]]))
    return nphi - gaps.max() + 1

tot = dict(hits=0, cone_box=0, shape_box=0, rows=0, rows_w=0, pairs=0, sampled_box=0, missed=0)
K, NEAR = 12, 0.2
for trial in range(60):
    f, i, j = rng.integers(6), rng.integers(N), rng.integers(N)
    Nv = face_point(f, np.float64((2 * i + 1) / N - 1), np.float64((2 * j + 1) / N - 1)); Nv /= np.linalg.norm(Nv)
    right = np.array([Nv[2], 0, -Nv[0]]); right /= np.linalg.norm(right); up = np.cross(Nv, right)
    st, ct = np.sin(thetas)[None, :], np.cos(thetas)[None, :]
    d = (st * np.cos(phis)[:, None])[..., None] * right + (st * np.sin(phis)[:, None])[..., None] * up + (ct * np.ones((nphi, 1)))[..., None] * Nv
    face, s, t = cube_face_st(d)
    i0 = np.clip(np.floor(s * ENV - 0.5).astype(int), 0, ENV - 1) >> 9; j0 = np.clip(np.floor(t * ENV - 0.5).astype(int), 0, ENV - 1) >> 9
    region = (j0 * 4 + i0) * 6 + face
    for r in range(96):
        hit = region == r
        rf, cell = r % 6, r // 6; rx, ry = cell % 4, cell // 4
        x0, x1, y0, y1 = rx * RS, (rx + 1) * RS, ry * RS, (ry + 1) * RS
        u0 = -1.0 if x0 == 0 else (2 * x0 + 1) / ENV - 1; u1 = 1.0 if x1 == ENV else (2 * x1 + 1) / ENV - 1
        v0 = -1.0 if y0 == 0 else (2 * y0 + 1) / ENV - 1; v1 = 1.0 if y1 == ENV else (2 * y1 + 1) / ENV - 1
        # box from K + 1 points on each edge of the region (theta, phi in the texel's frame)
        uu = np.concatenate([np.linspace(u0, u1, K + 1), np.linspace(u0, u1, K + 1), np.full(K + 1, u0), np.full(K + 1, u1)])
        vv = np.concatenate([np.full(K + 1, v0), np.full(K + 1, v1), np.linspace(v0, v1, K + 1), np.linspace(v0, v1, K + 1)])
        pts = face_point(rf, uu, vv)
        px, py, pz = pts @ right, pts @ up, pts @ Nv
        th = np.arctan2(np.hypot(px, py), pz)
        cc = face_point(rf, np.float64(0.5 * (u0 + u1)), np.float64(0.5 * (v0 + v1)))
        phc = np.arctan2(cc @ up, cc @ right)
        dl = np.arctan2(py, px) - phc; dl = (dl + np.pi) % (2 * np.pi) - np.pi
        fN, sN, tN = cube_face_st(Nv[None, :])
        inside = fN[0] == rf and u0 <= 2 * sN[0] - 1 <= u1 and v0 <= 2 * tN[0] - 1 <= v1
        thmin = 0.0 if inside else th.min()
        if not (not inside and thmin - 0.004 >= np.pi / 2):
            sb0 = max(0, int(np.floor((thmin - 0.004) / step)) - 1); sb1 = min(ntheta - 1, int(np.floor((th.max() + 0.004) / step)) + 1)
            ring = np.zeros(nphi, bool)
            if inside or thmin < NEAR:
                ring[:] = True
            else:
                lo, hi = phc + dl.min() - 0.004, phc + dl.max() + 0.004
                alo, ahi = int(np.floor(lo / step)) - 1, int(np.floor(hi / step)) + 1
                for a in range(alo, ahi + 1):
                    # phi index a stands for phi_a and phi_a +- 2 pi
                    for sh in (0, 1, -1):
                        q = a + int(round(sh * 2 * np.pi / step))
                        if 0 <= q < nphi: ring[q] = True
                # cells whose phi + 2 pi falls in the range
                for a in range(nphi):
                    for sh in (-2 * np.pi, 0, 2 * np.pi):
                        if lo - step <= phis[a] + sh <= hi + step: ring[a] = True
            box = np.zeros((nphi, ntheta), bool); box[np.ix_(ring, np.arange(sb0, sb1 + 1))] = True
            tot["sampled_box"] += int(box.sum()); tot["missed"] += int((hit & ~box).sum())
            # per phi row: the theta interval inside the four planes of the region (+ margin), within the box's rows
            H = [face_point(rf, np.float64(1), np.float64(0)) - face_point(rf, np.float64(0), np.float64(0)),
                 face_point(rf, np.float64(0), np.float64(1)) - face_point(rf, np.float64(0), np.float64(0)), face_point(rf, np.float64(0), np.float64(0))]
            es, et, em = H
            planes = [es - u0 * em, u1 * em - es, et - v0 * em, v1 * em - et]
            lo_t, hi_t = np.zeros(nphi), np.full(nphi, np.pi / 2)
            empty = np.zeros(nphi, bool)
            EPS = 0.01   # sine of the angular margin
            for h in planes:
                nh = np.linalg.norm(h)
                A = (h @ Nv) / nh; B = ((h @ right) * np.cos(phis) + (h @ up) * np.sin(phis)) / nh
                A = A + EPS; B = B + EPS   # relaxed: A cos + B sin >= -EPS (cos + sin) is implied by the exact condition
                both = (A < 0) & (B < 0); empty |= both
                c1 = (A >= 0) & (B < 0); hi_t = np.where(c1, np.minimum(hi_t, np.arctan2(A, -B)), hi_t)
                c2 = (A < 0) & (B >= 0); lo_t = np.where(c2, np.maximum(lo_t, np.arctan2(-A, B)), lo_t)
            rb0 = np.maximum(np.floor(lo_t / step).astype(int) - 1, sb0); rb1 = np.minimum(np.floor(hi_t / step).astype(int) + 1, sb1)
            cnt = np.where(empty | ~ring, 0, np.maximum(rb1 - rb0 + 1, 0))
            rowbox = np.zeros((nphi, ntheta), bool)
            for a in range(nphi):
                if cnt[a] > 0: rowbox[a, rb0[a]:rb1[a] + 1] = True
            tot["plane_rows"] = tot.get("plane_rows", 0) + int(rowbox.sum()); tot["plane_rows_missed"] = tot.get("plane_rows_missed", 0) + int((hit & ~rowbox).sum())
            tot["hybrid"] = tot.get("hybrid", 0) + int(rowbox.sum() if ring.all() else box.sum())
            key = "ring_full" if ring.all() else "ring_part"
            tot[key] = tot.get(key, 0) + int(box.sum()); tot[key + "_hits"] = tot.get(key + "_hits", 0) + int(hit.sum())
            if hit.any() and not ring.all():
                tot["part_ideal"] = tot.get("part_ideal", 0) + min(nphi, arc_of(hit) + 2) * (np.ptp(np.where(hit.any(0))[0]) + 3)
        else:
            tot["missed"] += int(hit.sum())
        # cone box (the kernel's)
        c = face_point(rf, np.float64(0.5 * (u0 + u1)), np.float64(0.5 * (v0 + v1))); c /= np.linalg.norm(c)
        corners = [face_point(rf, np.float64(a), np.float64(b)) for a in (u0, u1) for b in (v0, v1)]
        alpha = max(np.arccos(np.clip(np.dot(c, k / np.linalg.norm(k)), -1, 1)) for k in corners) + 0.004
        thc = np.arccos(np.clip(np.dot(c, Nv), -1, 1))
        if thc - alpha >= np.pi / 2:
            assert hit.sum() == 0
            continue
        tot["pairs"] += 1
        b0 = max(0, int(np.floor((thc - alpha) / step)) - 1); b1 = min(ntheta - 1, int(np.floor((thc + alpha) / step)) + 1)
        full = thc <= alpha or np.sin(alpha) >= 0.999 * np.sin(thc)
        if not full:
            dphi = np.arcsin(min(np.sin(alpha) / np.sin(thc), 1)) + 0.004
            na = min(nphi, int(np.floor(2 * dphi / step)) + 3)
        else:
            na = nphi
        tot["cone_box"] += na * (b1 - b0 + 1)
        # the same box, but phi from the azimuths of the region's corners when N lies outside the region, and theta's upper
        # end from the farthest corner
        fNn, sNn, tNn = cube_face_st(Nv[None, :])
        slack = 2.0 / ENV
        n_in = fNn[0] == rf and u0 - slack <= 2 * sNn[0] - 1 <= u1 + slack and v0 - slack <= 2 * tNn[0] - 1 <= v1 + slack
        cz = np.array([k @ Nv / np.linalg.norm(k) for k in corners]); cth = np.arccos(np.clip(cz, -1, 1))
        b1c = min(b1, int(np.floor((cth.max() + 0.004) / step)) + 1)
        na2 = na
        if not n_in:
            phc0 = np.arctan2(c @ up, c @ right)
            d = np.array([np.arctan2(k @ up, k @ right) for k in corners]) - phc0
            d = (d + np.pi) % (2 * np.pi) - np.pi
            na2 = min(na, int(np.floor((d.max() - d.min() + 0.008) / step)) + 4)
        tot["corner_box"] = tot.get("corner_box", 0) + na2 * (b1c - b0 + 1)
        if na == nphi:
            tot["full_pairs"] = tot.get("full_pairs", 0) + 1
            tot["full_pairs_inside"] = tot.get("full_pairs_inside", 0) + int(n_in)
            tot["full_cells"] = tot.get("full_cells", 0) + na * (b1 - b0 + 1)
            tot["full_cells_corner"] = tot.get("full_cells_corner", 0) + na2 * (b1c - b0 + 1)
        tot["hits"] += int(hit.sum())
        # bounding box of the true hit set + 1 index margin each side (what an ideal shape box would give)
        if hit.any():
            bs = np.where(hit.any(0))[0]; as_ = np.where(hit.any(1))[0]
            # phi range with wrap: smallest arc covering
            gaps = np.diff(np.concatenate([as_, [as_[0] + nphi]])); arc = nphi - gaps.max() + 1
            tot["shape_box"] += min(nphi, arc + 2) * min(ntheta, bs[-1] - bs[0] + 3)
            cnt = hit.sum(1)
            rows = cnt[cnt > 0] + 2
            tot["rows"] += int(rows.sum()) + 2 * (bs[-1] - bs[0] + 3)   # + one empty margin row either side
            W = 1 << int(np.ceil(np.log2(rows.max())))
            per = max(64 // W, 1)
            tot["rows_w"] += int(np.ceil((len(rows) + 2) / per)) * 64
h = tot["hits"]
print({k: (v, round(v / h, 2)) for k, v in tot.items()})
