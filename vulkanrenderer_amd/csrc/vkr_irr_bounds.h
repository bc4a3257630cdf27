// Which samples of irradiance.frag:21-36 can land in a rectangle of one cube face -- the culling arithmetic of the
// irradiance kernels that work through the environment region by region.  Everything in here is a BOUND: it may
// name samples that do not land in the rectangle (the kernels decide every named sample exactly, with the arithmetic
// of the lookup itself), it must never miss one that does.  The file compiles for the device and for the host, so
// that a host program can hold the bounds against an exhaustive classification (tests/cpp/irr_bounds_check.cpp).
//
// Setting.  An output texel has the tangent frame (right, up, N) of irradiance.frag:13-15 and samples the directions
//     d(a, b) = sin(theta_b) (cos(phi_a) right + sin(phi_a) up) + cos(theta_b) N,    phi_a ~ 0.025 a,  theta_b ~ 0.025 b.
// Frame and directions are given in the coordinates of the rectangle's face: (sc, tc, ma) as the cube-face selection
// defines them, ma positive on the face.  The rectangle is u0 <= sc/ma <= u1, v0 <= tc/ma <= v1 inside [-1, 1]^2.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VKR_HD __host__ __device__ __forceinline__
#else
#define VKR_HD static inline
#endif

namespace vkr {

struct irr_box { int a0, na, b0, nb; };  // phi indices a0 .. a0+na-1 (mod nphi), theta indices b0 .. b0+nb-1; na = 0: none

static constexpr float IRR_STEP = 0.025f;    // irradiance.frag:19, sampleDelta
static constexpr float IRR_MARGIN = 0.004f;  // radians by which every angular bound is widened
static constexpr float IRR_PI = 3.14159265359f;

struct irr_v3 { float x, y, z; };
VKR_HD irr_v3 irr_mk(float x, float y, float z) { irr_v3 r; r.x = x; r.y = y; r.z = z; return r; }
VKR_HD float irr_dot(irr_v3 a, irr_v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
VKR_HD irr_v3 irr_cross(irr_v3 a, irr_v3 b) { return irr_mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))); }
// Reciprocal, reciprocal root and root: bounds need them to a few ulps only, so the device takes the hardware's
// estimates (1 ulp) instead of the correctly rounded sequences; everything built on them is widened by IRR_MARGIN.
#if defined(__HIP_DEVICE_COMPILE__)
VKR_HD float irr_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
VKR_HD float irr_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
VKR_HD float irr_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
VKR_HD float irr_rcp(float x) { return 1.0f / x; }
VKR_HD float irr_rsq(float x) { return 1.0f / sqrtf(x); }
VKR_HD float irr_sqrt(float x) { return sqrtf(x); }
#endif
VKR_HD irr_v3 irr_unit(irr_v3 a) { const float s = irr_rsq(irr_dot(a, a)); return irr_mk(a.x * s, a.y * s, a.z * s); }

// atan on [0, 1] to 1e-5 (Abramowitz & Stegun 4.4.49)
VKR_HD float irr_atan_unit(float x)
{
    const float x2 = x * x;
    return x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 0.0208351f, -0.0851330f), 0.1801410f), -0.3302995f), 0.9998660f);
}
// atan on [0, inf] (reflected beyond 1) and atan2 in (-pi, pi], both to about 1e-5: enough for bounds that are widened
// by IRR_MARGIN afterwards
VKR_HD float irr_atan_pos(float t)
{
    const bool big = t > 1.0f;
    const float p = irr_atan_unit(big ? irr_rcp(t) : t);
    return big ? 0.5f * IRR_PI - p : p;
}
VKR_HD float irr_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float p = mx > 0.0f ? irr_atan_unit(mn * irr_rcp(mx)) : 0.0f;
    p = ay > ax ? 0.5f * IRR_PI - p : p;
    p = x < 0.0f ? IRR_PI - p : p;
    return y < 0.0f ? -p : p;
}
// sin on [0, 1.6] to 4e-6 (Taylor to x^9)
VKR_HD float irr_sin_small(float x)
{
    const float x2 = x * x;
    return x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -0.16666667f), 1.0f);
}
// the angle between two unit vectors, from the cross and the dot product: accurate near 0 and pi as well
VKR_HD float irr_angle(irr_v3 a, irr_v3 b)
{
    const irr_v3 c = irr_cross(a, b);
    return irr_atan2(irr_sqrt(irr_dot(c, c)), irr_dot(a, b));
}

// The box of (phi, theta) indices the rectangle spans as seen from N.  Polar angle: the farthest point of the
// rectangle is a corner, the nearest lies on an edge (an arc of a great circle) or is N itself.  Azimuth: if N lies
// outside, the rectangle spans less than half a turn, between the azimuths of two of its corners.  `slack` widens the
// rectangle for the test "N inside" (in units of sc/ma).  The angles come from the short approximations above
// (a few 1e-5 rad in total against the 4e-3 of IRR_MARGIN).
VKR_HD irr_box irr_pair_box(irr_v3 Nf, irr_v3 rightf, irr_v3 upf, float u0, float u1, float v0, float v1, float slack, int nphi, int ntheta)
{
    const float us[4] = {u0, u1, u0, u1}, vs[4] = {v0, v0, v1, v1};
    irr_v3 corner[4];
    float th_max = 0.0f, th_corner[4];
    const irr_v3 centre = irr_mk(0.5f * (u0 + u1), 0.5f * (v0 + v1), 1.0f);
    const float phc = irr_atan2(irr_dot(centre, upf), irr_dot(centre, rightf));
    float d_lo = 0.0f, d_hi = 0.0f;  // azimuth relative to the centre's
    for (int k = 0; k < 4; k++) {
        corner[k] = irr_unit(irr_mk(us[k], vs[k], 1.0f));
        th_corner[k] = irr_angle(corner[k], Nf);
        th_max = fmaxf(th_max, th_corner[k]);
        float d = irr_atan2(irr_dot(corner[k], upf), irr_dot(corner[k], rightf)) - phc;
        d = d > IRR_PI ? d - 2.0f * IRR_PI : (d <= -IRR_PI ? d + 2.0f * IRR_PI : d);
        d_lo = fminf(d_lo, d); d_hi = fmaxf(d_hi, d);
    }
    const bool n_inside = Nf.z > 0.0f && Nf.x >= (u0 - slack) * Nf.z && Nf.x <= (u1 + slack) * Nf.z && Nf.y >= (v0 - slack) * Nf.z && Nf.y <= (v1 + slack) * Nf.z;
    float th_min = 0.0f;
    if (!n_inside) {
        th_min = 4.0f;
        const int ea[4] = {0, 1, 3, 2}, eb[4] = {1, 3, 2, 0};
        for (int e = 0; e < 4; e++) {
            const irr_v3 A = corner[ea[e]], B = corner[eb[e]];
            const irr_v3 nrm = irr_unit(irr_cross(A, B));
            // the foot of the perpendicular from N lies on the arc when N is on the inner side of both end planes
            const bool on_arc = irr_dot(irr_cross(A, Nf), nrm) >= 0.0f && irr_dot(irr_cross(Nf, B), nrm) >= 0.0f;
            const float h = fminf(fabsf(irr_dot(Nf, nrm)), 1.0f);  // sine of the distance to the arc's great circle
            const float d = on_arc ? irr_atan2(h, irr_sqrt(fmaf(-h, h, 1.0f))) : fminf(th_corner[ea[e]], th_corner[eb[e]]);
            th_min = fminf(th_min, d);
        }
    }
    irr_box bx; bx.a0 = 0; bx.na = 0; bx.b0 = 0; bx.nb = 0;
    const float inv = 1.0f / IRR_STEP;
    if (!((th_min - IRR_MARGIN) * inv < (float)ntheta)) return bx;
    bx.b0 = (int)ceilf((th_min - IRR_MARGIN) * inv);
    bx.b0 = bx.b0 < 0 ? 0 : bx.b0;
    int b1 = (int)floorf((th_max + IRR_MARGIN) * inv);
    b1 = b1 > ntheta - 1 ? ntheta - 1 : b1;
    bx.nb = b1 - bx.b0 + 1;
    // IRR_MARGIN of arc is this much azimuth at the polar angle where the rectangle comes nearest to N
    const float margin = IRR_MARGIN * irr_rcp(fmaxf(irr_sin_small(fminf(th_min, 1.6f)), 1e-6f));
    bool full = n_inside || (d_hi - d_lo) + 2.0f * margin >= 2.0f * IRR_PI - 2.0f * IRR_STEP;
    if (!full) {
        const float centre_phi = phc < 0.0f ? phc + 2.0f * IRR_PI : phc;
        const float lo = centre_phi + d_lo - margin, hi = centre_phi + d_hi + margin;
        // phi index 0 also stands for the angle 2 pi, index nphi - 1 for just below it
        const int alo = (int)ceilf(lo * inv), ahi = (int)floorf(hi * inv);
        if (lo < 0.0f && hi >= 2.0f * IRR_PI) full = true;
        else if (lo < 0.0f) {  // phi in [lo + 2 pi, 2 pi) and in [0, hi]
            int w0 = (int)ceilf((lo + 2.0f * IRR_PI) * inv);
            w0 = w0 > nphi ? nphi : w0;
            if (w0 <= ahi + 1) full = true;
            else { bx.a0 = w0; bx.na = (nphi - w0) + (ahi + 1); }
        } else if (hi >= 2.0f * IRR_PI) {  // phi in [lo, 2 pi) and in [0, hi - 2 pi]
            const int e = (int)floorf((hi - 2.0f * IRR_PI) * inv), first = alo > nphi ? nphi : alo;
            if (e >= first - 1) full = true;
            else { bx.a0 = first; bx.na = (nphi - first) + (e + 1); }
        } else {
            const int last = ahi > nphi - 1 ? nphi - 1 : ahi;
            bx.a0 = alo; bx.na = last - alo + 1 > 0 ? last - alo + 1 : 0;
        }
    }
    if (full || bx.na >= nphi) { bx.a0 = 0; bx.na = nphi; }
    if (bx.nb <= 0) { bx.na = 0; bx.nb = 0; }
    return bx;
}

// r in the coordinates of face f: (sc, tc) as the cube-face selection gives them on that face, and the major axis
// signed so that it is positive on the face.  Components are swapped and negated only, which commutes with rounding.
VKR_HD irr_v3 irr_to_face_axes(int f, irr_v3 r)
{
    switch (f) {
    case 0: return irr_mk(-r.z, -r.y, r.x);
    case 1: return irr_mk(r.z, -r.y, -r.x);
    case 2: return irr_mk(r.x, r.z, r.y);
    case 3: return irr_mk(r.x, -r.z, -r.y);
    case 4: return irr_mk(r.x, -r.y, r.z);
    default: return irr_mk(-r.x, -r.y, -r.z);
    }
}

// Regions of the by-region order with row sums: the base texel of a lookup's footprint, shifted by one, is
// (I0, J0) in [0, n]^2; cells of 3 x 3 such positions, `cells_per_region` cells per region side (as
// cube_region_grid3 of the oracle).  Region (rx, ry) holds I0 in [X_lo, X_hi), i.e. the unrounded texel coordinate
// x = s n - 1/2 in [X_lo - 1, X_hi - 1); the first and the last region of a side also take what clamping sends them.
struct irr_region_rect {
    float x_lo, x_hi, y_lo, y_hi;  // exact membership: x_lo <= x < x_hi, y_lo <= y < y_hi
    float u0, u1, v0, v1;          // the rectangle in sc/ma, tc/ma, clamped to the face
    float U0, U1, V0, V1;          // widened by a texel, for irr_row_interval
};
VKR_HD void irr_region_grid(int n, int *per_side, int *cells_per_region)
{
#ifndef VKR_IRR_REGION_CELLS
#define VKR_IRR_REGION_CELLS 128
#endif
    const int cells = n / 3 + 1;
    *per_side = (cells + VKR_IRR_REGION_CELLS - 1) / VKR_IRR_REGION_CELLS;
    *cells_per_region = (cells + *per_side - 1) / *per_side;
}
VKR_HD irr_region_rect irr_region_geometry(int n, int per_side, int cells_per_region, int rx, int ry)
{
    irr_region_rect g;
    const float rn = 1.0f / (float)n;
    const int X_lo = 3 * cells_per_region * rx, Y_lo = 3 * cells_per_region * ry;
    int X_hi = 3 * cells_per_region * (rx + 1), Y_hi = 3 * cells_per_region * (ry + 1);
    X_hi = X_hi > n + 1 ? n + 1 : X_hi; Y_hi = Y_hi > n + 1 ? n + 1 : Y_hi;
    g.x_lo = rx == 0 ? -INFINITY : (float)(X_lo - 1); g.x_hi = rx == per_side - 1 ? INFINITY : (float)(X_hi - 1);
    g.y_lo = ry == 0 ? -INFINITY : (float)(Y_lo - 1); g.y_hi = ry == per_side - 1 ? INFINITY : (float)(Y_hi - 1);
    g.u0 = fmaxf((float)(2 * X_lo - 1) * rn - 1.0f, -1.0f); g.u1 = fminf((float)(2 * X_hi - 1) * rn - 1.0f, 1.0f);
    g.v0 = fmaxf((float)(2 * Y_lo - 1) * rn - 1.0f, -1.0f); g.v1 = fminf((float)(2 * Y_hi - 1) * rn - 1.0f, 1.0f);
    const float w = 2.0f * rn;
    g.U0 = g.u0 - w; g.U1 = g.u1 + w; g.V0 = g.v0 - w; g.V1 = g.v1 + w;
    return g;
}

// The theta indices of phi row (cp, sp) = (cos phi_a, sin phi_a) whose samples can land in the rectangle
// U0 <= sc/ma <= U1, V0 <= tc/ma <= V1 (the caller passes the rectangle widened by what its own coordinate
// rounding can amount to).  Along the row d = sin(theta) e + cos(theta) N with e = cp right + sp up, and each side
// of the rectangle is a plane g . d >= 0 through the origin: tan(theta) (g . e) + g . N >= 0, a half line in
// tan(theta).  [*bl, *bh] is the intersection of the four, widened by IRR_MARGIN; empty when *bl > *bh.
VKR_HD void irr_row_interval(irr_v3 Nf, irr_v3 rightf, irr_v3 upf, float cp, float sp, float U0, float U1, float V0, float V1, int *bl, int *bh)
{
    const float ex = fmaf(sp, upf.x, cp * rightf.x), ey = fmaf(sp, upf.y, cp * rightf.y), ez = fmaf(sp, upf.z, cp * rightf.z);
    const float A[4] = {fmaf(-U0, ez, ex), fmaf(U1, ez, -ex), fmaf(-V0, ez, ey), fmaf(V1, ez, -ey)};
    const float B[4] = {fmaf(-U0, Nf.z, Nf.x), fmaf(U1, Nf.z, -Nf.x), fmaf(-V0, Nf.z, Nf.y), fmaf(V1, Nf.z, -Nf.y)};
    float lo = 0.0f, hi = INFINITY;
    bool none = false;
    for (int k = 0; k < 4; k++) {
        // NaN (0 x inf) is dropped by fmaxf / fminf: no constraint; a relative error of 2^-22 in tan(theta) moves theta by
        // less than 2^-22 rad
        const float q = -B[k] * irr_rcp(A[k]);
        if (A[k] > 0.0f) lo = fmaxf(lo, q);
        else if (A[k] < 0.0f) hi = fminf(hi, q);
        else if (B[k] < 0.0f) none = true;
    }
    const float inv = 1.0f / IRR_STEP;
    int l = (int)ceilf((irr_atan_pos(lo) - IRR_MARGIN) * inv), h = hi < 0.0f ? -1 : (int)floorf((irr_atan_pos(hi) + IRR_MARGIN) * inv);
    if (none || !(lo <= hi)) { l = 1; h = 0; }
    *bl = l; *bh = h;
}

}  // namespace vkr
