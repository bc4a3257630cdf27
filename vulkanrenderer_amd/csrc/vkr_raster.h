// Pieces shared by the kernels that rasterise meshes (shadow maps, geometry pass): the matrix
// products of the vertex shaders in the library's fixed binary32 operation order, and the
// homogeneous binary64 coverage rule -- with V0,V1,V2 the clip-space (x,y,w) of a triangle's
// corners and p = (xn, yn, 1) a sample position in normalised device coordinates,
//   E0 = p.(V1xV2)  E1 = p.(V2xV0)  E2 = p.(V0xV1)  D = V0.(V1xV2);
// D > 0 is a back face under the reference's CCW-front convention, D < 0 a front face; the sample
// is inside (and in front of the eye) when E0,E1,E2 and their sum have the sign of D, and its
// depth passes the 0 <= z <= w clip when sum(Ei*zi) lies between 0 and D.  Attributes interpolate
// perspective-correctly as sum(Ei*ai) * (1 / sum(Ei)), and the depth is sum(Ei*zi) * (1 / D): one binary64
// division per sample or per triangle, then multiplications.
#pragma once
#include "vkr_device.h"

namespace vkr {

struct STri { double a[3], b[3], c[3], z[3], D, inv_D; };  // inv_D = 1 / D where the caller asks for it (raster_boxes<.., true>)

VKR_DEV double s_edge(const STri &t, int i, double xn, double yn) { return __builtin_fma(t.a[i], xn, __builtin_fma(t.b[i], yn, t.c[i])); }

// row-major 4x4 product in the row-vector convention, each element one fma chain
VKR_DEV float mat4_mul_elem(const float *a, const float *b, int i, int j)
{
    return fma_(a[4 * i + 3], b[12 + j], fma_(a[4 * i + 2], b[8 + j], fma_(a[4 * i + 1], b[4 + j], a[4 * i] * b[j])));
}

// GLSL mat4 * vec4 on row-major bytes: out[j] = sum_i v[i] * m[4i + j]
VKR_DEV void mat4_vec4(const float *m, const float (&v)[4], float (&o)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = fma_(v[3], m[12 + j], fma_(v[2], m[8 + j], fma_(v[1], m[4 + j], v[0] * m[j])));
}

VKR_DEV void raster_setup(const float (&clip)[3][4], STri &t)
{
    const double x0 = clip[0][0], y0 = clip[0][1], w0 = clip[0][3], x1 = clip[1][0], y1 = clip[1][1], w1 = clip[1][3], x2 = clip[2][0], y2 = clip[2][1], w2 = clip[2][3];
    t.a[0] = y1 * w2 - w1 * y2; t.b[0] = w1 * x2 - x1 * w2; t.c[0] = x1 * y2 - y1 * x2;
    t.a[1] = y2 * w0 - w2 * y0; t.b[1] = w2 * x0 - x2 * w0; t.c[1] = x2 * y0 - y2 * x0;
    t.a[2] = y0 * w1 - w0 * y1; t.b[2] = w0 * x1 - x0 * w1; t.c[2] = x0 * y1 - y0 * x1;
    t.z[0] = clip[0][2]; t.z[1] = clip[1][2]; t.z[2] = clip[2][2];
    t.D = __builtin_fma(w0, t.c[0], __builtin_fma(y0, t.b[0], x0 * t.a[0]));
}

// Turns a front face (D < 0) into the same triangle with D > 0, exactly (negation is exact).
VKR_DEV void raster_flip(STri &t)
{
#pragma unroll
    for (int i = 0; i < 3; i++) { t.a[i] = -t.a[i]; t.b[i] = -t.b[i]; t.c[i] = -t.c[i]; }
    t.D = -t.D;
}

static constexpr double RASTER_SLACK = 1.0 / 1024.0;

// Conservative sample bounding box of the part of a clip-space triangle with z >= 0 and w > 0, on a grid of res_x x res_y samples.
VKR_DEV bool raster_bbox(const float (&c)[3][4], int res_x, int res_y, int &bx0, int &by0, int &bx1, int &by1)
{
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    int n = 0;
    bool unbounded = false;
    auto add = [&](double x, double y, double w) {
        n++;
        if (!(w > 1e-12)) { unbounded = true; return; }
        const double px = x / w, py = y / w;
        xmin = px < xmin ? px : xmin; xmax = px > xmax ? px : xmax;
        ymin = py < ymin ? py : ymin; ymax = py > ymax ? py : ymax;
    };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float *a = c[i], *b = c[(i + 1) % 3];
        const bool ain = a[2] >= 0.0f && a[3] > 0.0f, bin = b[2] >= 0.0f && b[3] > 0.0f;
        if (ain) add(a[0], a[1], a[3]);
        if (ain != bin) {
            const double ta = a[2], tb = b[2];
            double t = (ta != tb) ? ta / (ta - tb) : 0.5;
            if (!(t >= 0.0 && t <= 1.0)) { const double wa = a[3], wb = b[3]; t = (wa != wb) ? wa / (wa - wb) : 0.5; }
            add((double)a[0] + t * ((double)b[0] - (double)a[0]), (double)a[1] + t * ((double)b[1] - (double)a[1]), (double)a[3] + t * ((double)b[3] - (double)a[3]));
        }
    }
    if (n == 0) return false;
    if (unbounded) { xmin = ymin = -1e300; xmax = ymax = 1e300; }
    // Sample i has its centre at i + 0.5 texels: those with xmin <= centre <= xmax, widened by RASTER_SLACK texels for the
    // rounding of the corners' positions here and of the edge functions' coefficients (both of the order of 2^-50 of a
    // face's width, so far less).  A triangle that reaches no sample centre gets an empty box.
    const double half_x = 0.5 * (double)res_x, half_y = 0.5 * (double)res_y;
    double fx0 = __builtin_ceil((xmin + 1.0) * half_x - (0.5 + RASTER_SLACK)), fx1 = __builtin_floor((xmax + 1.0) * half_x - (0.5 - RASTER_SLACK));
    double fy0 = __builtin_ceil((ymin + 1.0) * half_y - (0.5 + RASTER_SLACK)), fy1 = __builtin_floor((ymax + 1.0) * half_y - (0.5 - RASTER_SLACK));
    if (!(fx0 > 0.0)) fx0 = 0.0;
    if (!(fy0 > 0.0)) fy0 = 0.0;
    if (!(fx1 < (double)(res_x - 1))) fx1 = (double)(res_x - 1);
    if (!(fy1 < (double)(res_y - 1))) fy1 = (double)(res_y - 1);
    if (fx0 > fx1 || fy0 > fy1) return false;
    bx0 = (int)fx0; by0 = (int)fy0; bx1 = (int)fx1; by1 = (int)fy1;
    return true;
}

VKR_DEV bool shadow_bbox(const float (&c)[3][4], int res, int &bx0, int &by0, int &bx1, int &by1) { return raster_bbox(c, res, res, bx0, by0, bx1, by1); }

// True when the triangle lies entirely outside one plane of the clip volume (-w <= x,y <= w, z >= 0):
// nothing of it can be covered, and the binary64 setup can be skipped.  Exact comparisons on the
// binary32 clip coordinates, so this only ever drops triangles the full test would drop as well.
VKR_DEV bool clip_reject(const float (&c)[3][4])
{
    bool xp = true, xn = true, yp = true, yn = true, zn = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        xp = xp && c[k][0] > c[k][3]; xn = xn && c[k][0] < -c[k][3];
        yp = yp && c[k][1] > c[k][3]; yn = yn && c[k][1] < -c[k][3];
        zn = zn && c[k][2] < 0.0f;
    }
    return xp || xn || yp || yn || zn;
}

VKR_DEV float readlane_f(float x, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); }

#ifndef VKR_RASTER_BIG
#define VKR_RASTER_BIG 16
#endif
static constexpr int RASTER_BIG = VKR_RASTER_BIG;  // bounding boxes with more samples than this are walked by the whole wave (0 to 32 measured equal on a scene of boxes of about a hundred samples, 64 25 % slower)

// One wave's share of LDS for raster_boxes: the set-up triangles of the lanes whose boxes the whole wave walks.
struct alignas(16) RasterRecord {
    double a[3], b[3], c[3], D, inv_D;
    float z[3]; unsigned tag;
    int bx0, by0; unsigned w, samples;
    float attr[9]; float inv_w;  // the first three components of the three attribute rows; 1 / w rounded
};
static_assert(sizeof(RasterRecord) == 160, "ten 16-byte LDS accesses per record");
struct RasterPool { RasterRecord rec[64]; };

VKR_DEV void raster_wave_fence()
{
    // LDS accesses of one wave are carried out in order; the fences keep the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Calls pixel(tri, attr, tag, i, j) for every sample (i, j) of the bounding box of each lane's
// triangle (`have`: the lane has one; of attr's fourth components pixel sees zeros).  A lane walks a small box itself;
// a larger box is taken over by the whole wave, every lane taking every 64th sample, so one large triangle does not
// leave 63 lanes idle: each such triangle is set up by its own lane (all of them at once), left in the wave's `pool`
// in LDS, and read from there by all lanes when its turn comes.  (Before, its corners were broadcast and every lane
// redid the set-up, which cost more than walking a box of a hundred samples.)
// Every lane of the wave must call this (no early returns before it), with the wave's own pool.  FLIP: front faces were kept.
template <bool FLIP, bool INV_D, typename F>
VKR_DEV void raster_boxes(RasterPool &pool, bool have, const float (&clip)[3][4], const float (&attr)[3][4], unsigned tag,
                          int bx0, int by0, int bx1, int by1, F pixel)
{
    const int w = have ? bx1 - bx0 + 1 : 0, h = have ? by1 - by0 + 1 : 0;
    const bool big = have && (long long)w * h > RASTER_BIG;
    STri mine;
    if (have) {
        raster_setup(clip, mine);
        if (FLIP) raster_flip(mine);
        mine.inv_D = INV_D ? 1.0 / mine.D : 0.0;
    }
    if (have && !big) {
        for (int j = by0; j <= by1; j++)
            for (int i = bx0; i <= bx1; i++) pixel(mine, attr, tag, i, j);
    }
    unsigned long long todo = __ballot(big);
    if (todo == 0ull) return;  // wave-uniform
    const int lane = (int)__lane_id();
    if (big) {
        RasterRecord &r = pool.rec[lane];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r.a[k] = mine.a[k]; r.b[k] = mine.b[k]; r.c[k] = mine.c[k]; r.z[k] = clip[k][2];
#pragma unroll
            for (int j = 0; j < 3; j++) r.attr[3 * k + j] = attr[k][j];
        }
        r.D = mine.D; r.inv_D = mine.inv_D; r.tag = tag;
        r.bx0 = bx0; r.by0 = by0; r.w = (unsigned)w; r.samples = (unsigned)w * (unsigned)h;  // boxes hold fewer than 2^31 samples: faces and frames are at most 32768 wide
        r.inv_w = 1.0f / (float)w;
    }
    raster_wave_fence();
    while (todo) {  // wave-uniform
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const RasterRecord &r = pool.rec[src];
        STri t;
        float a2[3][4];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            t.a[c] = r.a[c]; t.b[c] = r.b[c]; t.c[c] = r.c[c]; t.z[c] = r.z[c];
#pragma unroll
            for (int j = 0; j < 3; j++) a2[c][j] = r.attr[3 * c + j];
            a2[c][3] = 0.0f;
        }
        t.D = r.D; t.inv_D = r.inv_D;
        const unsigned tag2 = r.tag;
        const int x0 = r.bx0, y0 = r.by0;
        // sample q = lane, lane + 64, ... of the box in row order; (ii, jj) advance by 64 samples without a division per step
        const unsigned uw = r.w, total = r.samples;
        const unsigned step_rows = (unsigned)(64.0f * r.inv_w);  // 64 / uw: the product is exact or within 2^-18 of a quotient that is not whole
        const unsigned step_cols = 64u - step_rows * uw;
        unsigned jj = (unsigned)((float)lane * r.inv_w), ii = (unsigned)lane - jj * uw;
        if ((int)ii < 0) { jj--; ii += uw; } else if (ii >= uw) { jj++; ii -= uw; }
        for (unsigned q = (unsigned)lane; q < total; q += 64u) {
            pixel(t, a2, tag2, x0 + (int)ii, y0 + (int)jj);
            ii += step_cols; jj += step_rows;
            if (ii >= uw) { ii -= uw; jj++; }
        }
    }
    raster_wave_fence();
}

// A stride coprime to n, near n / golden ratio: consecutive lanes then take triangles far apart in the mesh.
inline unsigned scatter_stride_for(unsigned n)
{
    if (n < 4) return 1;
    auto gcd = [](unsigned a, unsigned b) { while (b) { unsigned t = a % b; a = b; b = t; } return a; };
    unsigned s = (unsigned)((double)n * 0.6180339887498949);
    if (s < 1) s = 1;
    while (gcd(s, n) != 1) s++;
    return s;
}

}  // namespace vkr
