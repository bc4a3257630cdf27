// Device-side building blocks shared by the gfx950 kernels: binary16 packing, the fixed
// arithmetic rules (which GLSL operation maps to which IEEE sequence) and the software texture
// units that stand in for the Vulkan samplers of VulkanHelper.cs:123-220.
//
// Arithmetic rules (every kernel follows them so that results do not depend on how a pass is
// split over workgroups, strips or GPUs):
//   x / y        -> x * rcp_rn(y)            rcp_rn, sqrt_rn = correctly rounded
//   normalize(v) -> v * rcp_rn(sqrt_rn(dot(v,v)))
//   dot(a,b)     -> fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   pow(x,5)     -> ((x*x)*(x*x))*x
//   a*b + c      -> fma only where written as fmaf(); the build uses -ffp-contract=off
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vkr_fastmath.h"

#define VKR_DEV __device__ __forceinline__

namespace vkr {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

struct f3 { float x, y, z; };

VKR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VKR_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VKR_DEV float rcp_rn(float x) { return 1.0f / x; }             // hipcc: correctly rounded by default
VKR_DEV float sqrt_rn(float x) { return __builtin_sqrtf(x); }  // hipcc: correctly rounded by default
VKR_DEV float max0(float x) { return __builtin_fmaxf(x, 0.0f); }
VKR_DEV float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
VKR_DEV float dot3(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
VKR_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VKR_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VKR_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VKR_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VKR_DEV f3 normalize3(f3 a) { return scale3(a, rcp_rn(sqrt_rn(dot3(a, a)))); }
// Same bits as normalize3 for every input, through the short sqrt / reciprocal sequences of
// vkr_fastmath.h (which fall back to the IEEE ones outside their exact domain).
VKR_DEV f3 normalize3_fast(f3 a)
{
    float s, inv;
    sqrt_and_rcp_rn(dot3(a, a), s, inv);
    return scale3(a, inv);
}
VKR_DEV f3 cross3(f3 a, f3 b)
{
    return mk3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
VKR_DEV float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// ---- RGBA16F texel access -----------------------------------------------------------------

struct rgba { float r, g, b, a; };

VKR_DEV rgba unpack4(uint2 v)
{
    half4_t h = __builtin_bit_cast(half4_t, v);
    rgba o; o.r = (float)h.x; o.g = (float)h.y; o.b = (float)h.z; o.a = (float)h.w;
    return o;
}
// Hides how a value was computed from the instruction selector.  Without it hipcc folds
// "fma, then convert to binary16" into v_fma_mixlo_f16, which rounds once (straight to binary16)
// instead of twice (binary32, then binary16) and so changes the stored bits in rare cases.
VKR_DEV float pin(float x) { asm("" : "+v"(x)); return x; }
VKR_DEV _Float16 to_f16(float x) { return (_Float16)pin(x); }  // round to nearest even
VKR_DEV uint2 pack4(float r, float g, float b, float a)
{
    half4_t h; h.x = to_f16(r); h.y = to_f16(g); h.z = to_f16(b); h.w = to_f16(a);
    return __builtin_bit_cast(uint2, h);
}
VKR_DEV rgba load_rgba16f(const void *img, size_t texel) { return unpack4(((const uint2 *)img)[texel]); }
VKR_DEV float round_f16(float x) { return (float)to_f16(x); }

// ---- cube maps ------------------------------------------------------------------------------

// Face selection; ties resolve z over y over x.  Faces: +X,-X,+Y,-Y,+Z,-Z.
VKR_DEV int cube_face(f3 r, float &sc, float &tc, float &ma)
{
    float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
    if (az >= ax && az >= ay) {
        ma = az;
        if (r.z >= 0.0f) { sc = r.x; tc = -r.y; return 4; }
        sc = -r.x; tc = -r.y; return 5;
    }
    if (ay >= ax) {
        ma = ay;
        if (r.y >= 0.0f) { sc = r.x; tc = r.z; return 2; }
        sc = r.x; tc = -r.z; return 3;
    }
    ma = ax;
    if (r.x >= 0.0f) { sc = -r.z; tc = -r.y; return 0; }
    sc = r.z; tc = -r.y; return 1;
}

VKR_DEV void cube_st(f3 r, int &face, float &s, float &t)
{
    float sc, tc, ma;
    face = cube_face(r, sc, tc, ma);
    float ima = rcp_rn_checked(ma);
    s = fma_(sc * ima, 0.5f, 0.5f);
    t = fma_(tc * ima, 0.5f, 0.5f);
}

// Texel one step over a face edge -> the touching texel of the adjacent face (seamless cube
// filtering), on the integer lattice a = 2i+1-n.  Returns false at a cube corner.
VKR_DEV bool cube_fold(int n, int f, int i, int j, int &of, int &oi, int &oj)
{
    int a = 2 * i + 1 - n, b = 2 * j + 1 - n;
    bool aout = (a > n) || (a < -n), bout = (b > n) || (b < -n);
    if (!aout && !bout) { of = f; oi = i; oj = j; return true; }
    if (aout && bout) return false;
    int m = n - 1;
    if (aout) a = a > 0 ? n : -n;
    if (bout) b = b > 0 ? n : -n;
    int px, py, pz;
    switch (f) {
    case 0: px = m; py = -b; pz = -a; break;
    case 1: px = -m; py = -b; pz = a; break;
    case 2: px = a; py = m; pz = b; break;
    case 3: px = a; py = -m; pz = -b; break;
    case 4: px = a; py = -b; pz = m; break;
    default: px = -a; py = -b; pz = -m; break;
    }
    int sc, tc;
    if (pz == n) { of = 4; sc = px; tc = -py; }
    else if (pz == -n) { of = 5; sc = -px; tc = -py; }
    else if (py == n) { of = 2; sc = px; tc = pz; }
    else if (py == -n) { of = 3; sc = px; tc = -pz; }
    else if (px == n) { of = 0; sc = -pz; tc = -py; }
    else { of = 1; sc = pz; tc = -py; }
    oi = (sc + n - 1) / 2; oj = (tc + n - 1) / 2;
    return true;
}

// Index of texel (i, j) of face f in an RGBA16F cube of face size n.  Row-major, or (TILED)
// 4x4-texel tiles of 128 bytes stored row-major by tile (n % 4 == 0): the 2x2 footprint of a
// bilinear lookup then lies in one cache line 9 times out of 16 instead of never.
template <bool TILED> VKR_DEV size_t cube_texel_index(int n, int f, int i, int j)
{
    const size_t face0 = (size_t)f * n * n;
    if (!TILED) return face0 + (size_t)j * n + i;
    return face0 + (((unsigned)j >> 2) * ((unsigned)n >> 2) + ((unsigned)i >> 2)) * 16u + ((unsigned)j & 3u) * 4u + ((unsigned)i & 3u);
}

// Footprint of a LINEAR lookup in a cube of face size n: the face, the base texel (i0, j0) of the 2x2 texels
// (either may be -1 or n - 1: the footprint then reaches over a face edge) and the two interpolation weights.
struct cube_footprint { int face, i0, j0; float al, be; };
VKR_DEV cube_footprint cube_linear_footprint(int n, f3 r)
{
    cube_footprint p;
    float s, t;
    cube_st(r, p.face, s, t);
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    p.al = x - fx; p.be = y - fy;
    p.i0 = (int)fx; p.j0 = (int)fy;
    return p;
}

// The four texels of a footprint, weighted (seamless over face edges; at a cube corner the missing texel is the
// mean of the other three).
template <bool TILED>
VKR_DEV f3 texcube_fetch_footprint(const void *cube, int n, cube_footprint p)
{
    const int face = p.face, i0 = p.i0, j0 = p.j0;
    const float al = p.al, be = p.be;
    float w0 = (1.0f - al) * (1.0f - be), w1 = al * (1.0f - be), w2 = (1.0f - al) * be, w3 = al * be;
    const uint2 *base = (const uint2 *)cube;
    rgba t0, t1, t2, t3;
    if (i0 >= 0 && j0 >= 0 && i0 + 1 < n && j0 + 1 < n) {  // all four taps on this face
        if (!TILED) {
            const uint2 *p = base + ((size_t)face * n + j0) * n + i0;
            t0 = unpack4(p[0]); t1 = unpack4(p[1]); t2 = unpack4(p[n]); t3 = unpack4(p[n + 1]);
        } else {
            const uint2 *fb = base + (size_t)face * n * n;
            const unsigned tiles4 = (unsigned)n * 4u;  // texels in one row of tiles
            const unsigned c0 = ((unsigned)i0 >> 2) * 16u + ((unsigned)i0 & 3u), c1 = ((unsigned)(i0 + 1) >> 2) * 16u + ((unsigned)(i0 + 1) & 3u);
            const unsigned r0 = ((unsigned)j0 >> 2) * tiles4 + ((unsigned)j0 & 3u) * 4u, r1 = ((unsigned)(j0 + 1) >> 2) * tiles4 + ((unsigned)(j0 + 1) & 3u) * 4u;
            t0 = unpack4(fb[r0 + c0]); t1 = unpack4(fb[r0 + c1]); t2 = unpack4(fb[r1 + c0]); t3 = unpack4(fb[r1 + c1]);
        }
    } else {
        rgba tx[4];
        bool ok[4];
        int nok = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int of, oi, oj;
            ok[k] = cube_fold(n, face, i0 + (k & 1), j0 + (k >> 1), of, oi, oj);
            if (ok[k]) { tx[k] = unpack4(base[cube_texel_index<TILED>(n, of, oi, oj)]); nok++; }
            else { tx[k].r = tx[k].g = tx[k].b = tx[k].a = 0.0f; }
        }
        if (nok == 3) {  // cube corner: the missing texel is the mean of the other three
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) if (ok[k]) { sr += tx[k].r; sg += tx[k].g; sb += tx[k].b; }
#pragma unroll
            for (int k = 0; k < 4; k++) if (!ok[k]) { tx[k].r = sr * (1.0f / 3.0f); tx[k].g = sg * (1.0f / 3.0f); tx[k].b = sb * (1.0f / 3.0f); }
        }
        t0 = tx[0]; t1 = tx[1]; t2 = tx[2]; t3 = tx[3];
    }
    return mk3(fma_(w3, t3.r, fma_(w2, t2.r, fma_(w1, t1.r, w0 * t0.r))),
               fma_(w3, t3.g, fma_(w2, t2.g, fma_(w1, t1.g, w0 * t0.g))),
               fma_(w3, t3.b, fma_(w2, t2.b, fma_(w1, t1.b, w0 * t0.b))));
}

// LINEAR, seamless lookup in an RGBA16F cube, mip 0 (textureSampler on a one-level cube view)
template <bool TILED = false>
VKR_DEV f3 texcube_rgb_linear(const void *cube, int n, f3 r)
{
    return texcube_fetch_footprint<TILED>(cube, n, cube_linear_footprint(n, r));
}

VKR_DEV int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
// The same for a coordinate known to be within one period of [0, n) (anything else takes the division)
VKR_DEV int wrap_near(int i, int n)
{
    if (i < -n || i >= 2 * n) return wrap_repeat(i, n);
    return i < 0 ? i + n : (i >= n ? i - n : i);
}

// Point (s, t) in [-1, 1]^2 of face f as a direction (inverse of cube_face)
VKR_DEV f3 face_point(int f, float s, float t)
{
    switch (f) {
    case 0: return mk3(1.0f, -t, -s);
    case 1: return mk3(-1.0f, -t, s);
    case 2: return mk3(s, 1.0f, t);
    case 3: return mk3(s, -1.0f, -t);
    case 4: return mk3(s, -t, 1.0f);
    default: return mk3(-s, -t, -1.0f);
    }
}

// Direction through the centre of texel (i,j) of face f of an n^2 cube
VKR_DEV f3 face_dir(int f, int i, int j, int n)
{
    float rn = rcp_rn((float)n);
    float s = fma_((float)(2 * i + 1), rn, -1.0f), t = fma_((float)(2 * j + 1), rn, -1.0f);
    return face_point(f, s, t);
}

}  // namespace vkr
