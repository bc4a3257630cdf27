// Holds the culling bounds of the by-region irradiance kernel (vulkanrenderer_amd/csrc/vkr_irr_bounds.h) against an
// exhaustive classification on the host: for every output texel of an n^2 cube and every sample of
// irradiance.frag:21-36 the oracle says which region of an env_n^2 environment the lookup starts in
// (orc_cube_lookup_region3); every such sample must lie inside the (phi, theta) box of its (texel, region) pair and
// inside the theta interval of its phi row.  Prints how tight the bounds are, and how full the lanes of a wave would
// be when every lane walks one row's interval.  Exit status 1 if a sample is missed.
//
//   irr_bounds_check ENV_N CUBE_N [TEXEL_STRIDE]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../vulkanrenderer_amd/csrc/vkr_irr_bounds.h"

extern "C" int orc_cube_lookup_region3(int n, const float *dir3);

using namespace vkr;

static const float PI_F = 3.14159265359f;

static irr_v3 normalize3(irr_v3 a)
{
    const float s = 1.0f / sqrtf(irr_dot(a, a));
    return irr_mk(a.x * s, a.y * s, a.z * s);
}

static irr_v3 face_dir(int f, int i, int j, int n)
{
    const float rn = 1.0f / (float)n;
    const float s = fmaf((float)(2 * i + 1), rn, -1.0f), t = fmaf((float)(2 * j + 1), rn, -1.0f);
    switch (f) {
    case 0: return irr_mk(1.0f, -t, -s);
    case 1: return irr_mk(-1.0f, -t, s);
    case 2: return irr_mk(s, 1.0f, t);
    case 3: return irr_mk(s, -1.0f, -t);
    case 4: return irr_mk(s, -t, 1.0f);
    default: return irr_mk(-s, -t, -1.0f);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s ENV_N CUBE_N [TEXEL_STRIDE]\n", argv[0]); return 2; }
    const int env_n = atoi(argv[1]), n = atoi(argv[2]), stride = argc > 3 ? atoi(argv[3]) : 1;
    float phis[512], thetas[128];
    int nphi = 0, ntheta = 0;
    for (float phi = 0.0f; phi < 2.0f * PI_F; phi += 0.025f) phis[nphi++] = phi;
    for (float th = 0.0f; th < 0.5f * PI_F; th += 0.025f) thetas[ntheta++] = th;
    std::vector<float> cp(nphi), sp(nphi), st(ntheta), ct(ntheta);
    for (int a = 0; a < nphi; a++) { cp[a] = cosf(phis[a]); sp[a] = sinf(phis[a]); }
    for (int b = 0; b < ntheta; b++) { st[b] = sinf(thetas[b]); ct[b] = cosf(thetas[b]); }
    int per_side, cpr;
    irr_region_grid(env_n, &per_side, &cpr);
    const int n_regions = 6 * per_side * per_side;
    const int n_texels = 6 * n * n;

    long long missed = 0, hits = 0, cands = 0, box_cells = 0, row_tasks = 0, pairs = 0, wave_steps = 0, wave_count = 0, rows_with_hits = 0, skipped = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : missed, hits, cands, box_cells, row_tasks, pairs, wave_steps, wave_count, rows_with_hits, skipped)
    for (int texel = 0; texel < n_texels; texel += stride) {
        const int f = texel / (n * n), j = (texel / n) % n, i = texel % n;
        const irr_v3 N = normalize3(face_dir(f, i, j, n));
        const irr_v3 right = normalize3(irr_mk(N.z, 0.0f, -N.x));
        const irr_v3 up = normalize3(irr_cross(N, right));
        if (!(right.x == right.x) || !(up.x == up.x)) { skipped++; continue; }  // no tangent frame: the kernel sends every sample to region 1
        // per region: per phi row the first and last theta index that lands there, and the number that do
        std::vector<short> bmin((size_t)n_regions * nphi, 32767), bmax((size_t)n_regions * nphi, -1), cnt((size_t)n_regions * nphi, 0);
        std::vector<char> touched(n_regions, 0);
        for (int a = 0; a < nphi; a++)
            for (int b = 0; b < ntheta; b++) {
                const float tx = st[b] * cp[a], ty = st[b] * sp[a], tz = ct[b];
                const float d[3] = {fmaf(tz, N.x, fmaf(ty, up.x, tx * right.x)), fmaf(tz, N.y, fmaf(ty, up.y, tx * right.y)), fmaf(tz, N.z, fmaf(ty, up.z, tx * right.z))};
                const int r = orc_cube_lookup_region3(env_n, d);
                touched[r] = 1;
                const size_t k = (size_t)r * nphi + a;
                if (b < bmin[k]) bmin[k] = (short)b;
                if (b > bmax[k]) bmax[k] = (short)b;
                cnt[k]++;
            }
        std::vector<int> lens;  // interval lengths of this texel's row tasks, all regions: what consecutive lanes would hold
        for (int r = 0; r < n_regions; r++) {
            const int rf = r % 6, cell = r / 6, rx = cell % per_side, ry = cell / per_side;
            const irr_region_rect g = irr_region_geometry(env_n, per_side, cpr, rx, ry);
            const irr_v3 Nf = irr_to_face_axes(rf, N), rightf = irr_to_face_axes(rf, right), upf = irr_to_face_axes(rf, up);
            const irr_box bx = irr_pair_box(Nf, rightf, upf, g.u0, g.u1, g.v0, g.v1, 4.0f / (float)env_n, nphi, ntheta);
            std::vector<char> in_box(nphi, 0);
            if (bx.na > 0) {
                pairs++;
                box_cells += (long long)bx.na * bx.nb;
                for (int q = 0; q < bx.na; q++) {
                    const int a = (bx.a0 + q) % nphi;
                    in_box[a] = 1;
                    int bl, bh;
                    irr_row_interval(Nf, rightf, upf, cp[a], sp[a], g.U0, g.U1, g.V0, g.V1, &bl, &bh);
                    bl = bl < bx.b0 ? bx.b0 : bl;
                    bh = bh > bx.b0 + bx.nb - 1 ? bx.b0 + bx.nb - 1 : bh;
                    const int len = bh >= bl ? bh - bl + 1 : 0;
                    row_tasks++;
                    cands += len;
                    lens.push_back(len);
                    const size_t k = (size_t)r * nphi + a;
                    if (cnt[k] > 0) {
                        rows_with_hits++;
                        if (bmin[k] < bl || bmax[k] > bh) {
                            missed++;
                            if (missed < 5) fprintf(stderr, "texel %d region %d row %d: samples %d..%d, interval %d..%d (box b %d..%d)\n", texel, r, a, bmin[k], bmax[k], bl, bh, bx.b0, bx.b0 + bx.nb - 1);
                        }
                    }
                    hits += cnt[k];
                }
            }
            if (touched[r])
                for (int a = 0; a < nphi; a++)
                    if (cnt[(size_t)r * nphi + a] > 0 && !in_box[a]) {
                        missed++;
                        if (missed < 5) fprintf(stderr, "texel %d region %d: row %d holds samples but lies outside the box (a0 %d, na %d)\n", texel, r, a, bx.a0, bx.na);
                    }
        }
        for (size_t w = 0; w < lens.size(); w += 64) {
            int mx = 0;
            for (size_t k = w; k < lens.size() && k < w + 64; k++) mx = lens[k] > mx ? lens[k] : mx;
            wave_steps += mx;
            wave_count++;
        }
    }
    const long long expect = (long long)((n_texels + stride - 1) / stride - skipped) * nphi * ntheta;
    printf("environment %d^2 (%d regions of %d cells a side), cube %d^2, every %d. texel: %lld samples, %lld found in their rows' intervals, %lld rows or samples missed\n",
           env_n, n_regions, cpr, n, stride, expect, hits, missed);
    printf("  pairs with a box %lld, box cells per sample %.3f, interval cells per sample %.3f, row tasks %lld (%.2f samples per task, %.1f %% of tasks empty)\n",
           pairs, (double)box_cells / (double)hits, (double)cands / (double)hits, row_tasks, (double)hits / (double)row_tasks,
           100.0 * (double)(row_tasks - rows_with_hits) / (double)row_tasks);
    printf("  one row per lane: %lld waves, %.2f steps per wave, lanes busy %.1f %% of the steps, useful %.1f %%\n",
           wave_count, (double)wave_steps / (double)wave_count, 100.0 * (double)cands / (64.0 * (double)wave_steps), 100.0 * (double)hits / (64.0 * (double)wave_steps));
    return missed == 0 && hits == expect ? 0 : 1;
}
