/*
 * vkr_oracle.c -- CPU restatement of VulkanRenderer's screen-space shading path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls this file;
 * it exists so that tests, the smoke check and the benchmark's cpu_baseline leg have an
 * independent result to compare the HIP kernels with.
 *
 * Pinning.  The reference ships no tests, golden images or known-answer vectors for this path,
 * and neither its C# host code (needs .NET 8 + NuGet packages) nor a Vulkan driver is available
 * in the build container.  What the reference does ship is its compiled shaders
 * (the .frag.spv files under compiled_shaders/): tools/spirv_interp.py executes those binaries on small seeded
 * inputs with the reference's sampler / blend / attachment state, tools/make_golden.py stores the
 * inputs and outputs under tests/golden/, and tests/test_golden_reference_spirv.py holds this file
 * to them.  The fixed-function steps a Vulkan driver performs for the reference (texture
 * filtering, cube face selection, blending, binary16 conversion) are in no reference file; they are
 * restated from the Vulkan specification here and, independently, in the interpreter's samplers.
 * Closed-form known answers are in tests/test_oracle_kat.py.
 *
 * Source files restated (paths relative to the reference checkout):
 *   src/Renderer/Shaders/composition.frag:30-168      orc_shade (directional + ambient part)
 *   src/Renderer/Shaders/light.frag:35-141            orc_shade (per-light part)
 *   src/Renderer/GraphicsPipelineBuilder.cs:157-187   additive blend, applied per light draw
 *   src/Renderer/VulkanRenderer.cs:479-498            light draw order
 *   src/Renderer/Shaders/bloom_blur.frag:14-33        orc_bloom_blur
 *   src/Renderer/Shaders/postprocess.frag:11-38       orc_postprocess
 *   src/Renderer/Shaders/equirectangular_to_cubemap.frag:8-24   orc_equirect_to_cube
 *   src/Renderer/Shaders/irradiance.frag:10-39        orc_irradiance
 *   src/Renderer/Shaders/prefilter.frag:21-115        orc_prefilter
 *   src/Renderer/Shaders/brdf.frag:22-119             orc_brdf_lut
 *   src/Renderer/VulkanHelper.cs:123-220              the three samplers (filter / wrap / border)
 *   src/Renderer/VulkanRenderer.cs:656-665            cube face orientation
 *
 * Arithmetic profile.  GLSL leaves the precision of '/', sqrt, normalize, pow and of a*b+c
 * fusion to the implementation.  This restatement fixes one choice for each, so that a second
 * implementation following the same rules produces the same bits:
 *   - every value is IEEE binary32; evaluation order is the order of the GLSL source;
 *   - x / y            = x * RN(1/y)            (RN = correctly rounded)
 *   - sqrt             = correctly rounded
 *   - normalize(v)     = v * RN(1 / sqrt(dot(v,v)))
 *   - dot(a,b)         = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   - a product that is a direct operand of + or - is fused into that add (fma), as spelled
 *     out at each site below
 *   - pow(x, 5.0)      = ((x*x)*(x*x))*x
 *   - sin, cos, atan, asin, pow(x, 1/2.2) = the C library's float functions
 *   - attachment stores round to binary16 with round-to-nearest-even.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static const float PI_F = 3.14159265359f;

/* ------------------------------------------------------------------------------------------ */
/* binary16 <-> binary32                                                                        */

static inline float h2f(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline uint16_t f2h(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0u));
    if (ax >= 0x477FF000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    if (ax < 0x33000001u) /* <= 2^-25: rounds to zero (tie at exactly 2^-25 goes to even = 0) */
        return (uint16_t)sign;
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u; /* 24-bit significand */
    uint32_t shift, hbits;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hbits = 0;
    } else {
        shift = 13;
        hbits = (uint32_t)(e + 15) << 10;
        m &= 0x7FFFFFu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u)))
        q++;
    return (uint16_t)(sign | (hbits + q)); /* a carry out of the mantissa bumps the exponent */
}

ORC_API void orc_f32_to_f16(const float *src, uint16_t *dst, int64_t n)
{
    for (int64_t i = 0; i < n; i++) dst[i] = f2h(src[i]);
}

ORC_API void orc_f16_to_f32(const uint16_t *src, float *dst, int64_t n)
{
    for (int64_t i = 0; i < n; i++) dst[i] = h2f(src[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* small vector helpers following the arithmetic profile                                        */

typedef struct { float x, y, z; } v3;

static inline float rcpf_(float x) { return 1.0f / x; }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 v3_(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 sub3(v3 a, v3 b) { return v3_(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add3(v3 a, v3 b) { return v3_(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 scale3(v3 a, float s) { return v3_(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return v3_(-a.x, -a.y, -a.z); }
static inline v3 normalize3(v3 a) { return scale3(a, rcpf_(sqrtf(dot3(a, a)))); }
/* cross(a,b).x = a.y*b.z - a.z*b.y, fused as fma(a.y, b.z, -(a.z*b.y)) */
static inline v3 cross3(v3 a, v3 b)
{
    return v3_(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* ------------------------------------------------------------------------------------------ */
/* software texture units (Vulkan image-operation rules; sampler state from VulkanHelper.cs)    */

enum { ORC_WRAP_REPEAT = 0, ORC_WRAP_CLAMP = 1 };

static inline int wrap_i(int i, int n, int mode)
{
    if (mode == ORC_WRAP_REPEAT) {
        int m = i % n;
        return m < 0 ? m + n : m;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

/* float -> int as the GPU's conversion instruction defines it (C leaves NaN and out-of-range values
 * undefined): NaN gives 0, everything else saturates.  Only non-finite inputs ever get here with such values. */
static inline int f2i_sat(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int)x;
}

/* shadowSampler (VulkanHelper.cs:189-220): NEAREST, CLAMP_TO_BORDER, border = opaque white (1.0) */
static inline float tex2d_depth_nearest_border(const float *img, int res, float s, float t)
{
    float u = s * (float)res, v = t * (float)res;
    float fi = floorf(u), fj = floorf(v);
    if (!(fi >= 0.0f && fi < (float)res && fj >= 0.0f && fj < (float)res))
        return 1.0f;
    return img[(int)fj * res + (int)fi];
}

/* Cube-map face selection (Vulkan spec "Cube Map Face Selection"; ties: z over y over x).
 * Faces are stored +X,-X,+Y,-Y,+Z,-Z (SCDevice.cs:198-222 copies face f to layer f). */
static inline int cube_face(v3 r, float *sc, float *tc, float *ma)
{
    float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    if (az >= ax && az >= ay) {
        *ma = az;
        if (r.z >= 0.0f) { *sc = r.x;  *tc = -r.y; return 4; }
        *sc = -r.x; *tc = -r.y; return 5;
    }
    if (ay >= ax) {
        *ma = ay;
        if (r.y >= 0.0f) { *sc = r.x; *tc = r.z; return 2; }
        *sc = r.x; *tc = -r.z; return 3;
    }
    *ma = ax;
    if (r.x >= 0.0f) { *sc = -r.z; *tc = -r.y; return 0; }
    *sc = r.z; *tc = -r.y; return 1;
}

/* face coordinate in [0,1]: s = 1/2 * sc/|ma| + 1/2 */
static inline void cube_st(v3 r, int *face, float *s, float *t)
{
    float sc, tc, ma;
    *face = cube_face(r, &sc, &tc, &ma);
    float ima = rcpf_(ma);
    *s = fmaf(sc * ima, 0.5f, 0.5f);
    *t = fmaf(tc * ima, 0.5f, 0.5f);
}

/* shadowSampler on a depth cube: NEAREST (light.frag:133) */
static inline float texcube_depth_nearest(const float *cube, int res, v3 r)
{
    int face; float s, t;
    cube_st(r, &face, &s, &t);
    int i = f2i_sat(floorf(s * (float)res)), j = f2i_sat(floorf(t * (float)res));
    i = i < 0 ? 0 : (i >= res ? res - 1 : i);
    j = j < 0 ? 0 : (j >= res ? res - 1 : j);
    return cube[((size_t)face * res + j) * res + i];
}

/* Texel (i,j) of face f, where i or j may be -1 or n (one step over an edge), resolved to the
 * texel of the adjacent face that touches the same edge at the same position along it
 * (seamless cube filtering).  Works on the integer lattice a = 2i+1-n, so it is exact.
 * Returns 0 at a cube corner (both coordinates outside: no such texel). */
static int cube_fold(int n, int f, int i, int j, int *of, int *oi, int *oj)
{
    int a = 2 * i + 1 - n, b = 2 * j + 1 - n;
    int aout = (a > n) || (a < -n), bout = (b > n) || (b < -n);
    if (!aout && !bout) { *of = f; *oi = i; *oj = j; return 1; }
    if (aout && bout) return 0;
    int m = n; /* major-axis coordinate of the unfolded point, scaled by n */
    if (aout) { a = a > 0 ? n : -n; m = n - 1; }
    if (bout) { b = b > 0 ? n : -n; m = n - 1; }
    int p[3];
    switch (f) {
    case 0: p[0] = m;  p[1] = -b; p[2] = -a; break;
    case 1: p[0] = -m; p[1] = -b; p[2] = a;  break;
    case 2: p[0] = a;  p[1] = m;  p[2] = b;  break;
    case 3: p[0] = a;  p[1] = -m; p[2] = -b; break;
    case 4: p[0] = a;  p[1] = -b; p[2] = m;  break;
    default: p[0] = -a; p[1] = -b; p[2] = -m; break;
    }
    int nf, sc, tc;
    if (p[2] == n)       { nf = 4; sc = p[0];  tc = -p[1]; }
    else if (p[2] == -n) { nf = 5; sc = -p[0]; tc = -p[1]; }
    else if (p[1] == n)  { nf = 2; sc = p[0];  tc = p[2]; }
    else if (p[1] == -n) { nf = 3; sc = p[0];  tc = -p[2]; }
    else if (p[0] == n)  { nf = 0; sc = -p[2]; tc = -p[1]; }
    else                 { nf = 1; sc = p[2];  tc = -p[1]; }
    *of = nf; *oi = (sc + n - 1) / 2; *oj = (tc + n - 1) / 2;
    return 1;
}

ORC_API int orc_cube_fold(int n, int f, int i, int j, int *out3)
{
    return cube_fold(n, f, i, j, &out3[0], &out3[1], &out3[2]);
}

/* textureSampler on an RGBA16F cube: LINEAR, seamless, mip 0 only (sampler MaxLod = 0 and the
 * cube view has one level: VulkanHelper.cs:145,412-413) */
static inline v3 texcube_rgb_linear(const uint16_t *cube, int n, v3 r)
{
    int face; float s, t;
    cube_st(r, &face, &s, &t);
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float al = x - fx, be = y - fy;
    int i0 = f2i_sat(fx), j0 = f2i_sat(fy);
    float w[4] = {(1.0f - al) * (1.0f - be), al * (1.0f - be), (1.0f - al) * be, al * be};
    float tx[4][3];
    int ok[4], nok = 0;
    for (int k = 0; k < 4; k++) {
        int of, oi, oj;
        ok[k] = cube_fold(n, face, i0 + (k & 1), j0 + (k >> 1), &of, &oi, &oj);
        if (ok[k]) {
            const uint16_t *p = cube + (((size_t)of * n + oj) * n + oi) * 4;
            tx[k][0] = h2f(p[0]); tx[k][1] = h2f(p[1]); tx[k][2] = h2f(p[2]);
            nok++;
        }
    }
    if (nok == 3) { /* cube corner: the missing texel is the average of the other three */
        for (int k = 0; k < 4; k++) if (!ok[k]) {
            for (int c = 0; c < 3; c++) {
                float sum = 0.0f;
                for (int q = 0; q < 4; q++) if (ok[q]) sum += tx[q][c];
                tx[k][c] = sum * (1.0f / 3.0f);
            }
        }
    }
    float o[3];
    for (int c = 0; c < 3; c++)
        o[c] = fmaf(w[3], tx[3][c], fmaf(w[2], tx[2][c], fmaf(w[1], tx[1][c], w[0] * tx[0][c])));
    return v3_(o[0], o[1], o[2]);
}

/* textureSampler on a 2D RG16F image (the BRDF LUT): LINEAR, REPEAT on both axes */
static inline void tex2d_rg_linear_repeat(const uint16_t *img, int w, int h, float s, float t, float *o0, float *o1)
{
    float x = s * (float)w - 0.5f, y = t * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float al = x - fx, be = y - fy;
    int i0 = wrap_i((int)fx, w, ORC_WRAP_REPEAT), i1 = wrap_i((int)fx + 1, w, ORC_WRAP_REPEAT);
    int j0 = wrap_i((int)fy, h, ORC_WRAP_REPEAT), j1 = wrap_i((int)fy + 1, h, ORC_WRAP_REPEAT);
    float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
    const uint16_t *p00 = img + ((size_t)j0 * w + i0) * 2, *p10 = img + ((size_t)j0 * w + i1) * 2;
    const uint16_t *p01 = img + ((size_t)j1 * w + i0) * 2, *p11 = img + ((size_t)j1 * w + i1) * 2;
    *o0 = fmaf(w11, h2f(p11[0]), fmaf(w01, h2f(p01[0]), fmaf(w10, h2f(p10[0]), w00 * h2f(p00[0]))));
    *o1 = fmaf(w11, h2f(p11[1]), fmaf(w01, h2f(p01[1]), fmaf(w10, h2f(p10[1]), w00 * h2f(p00[1]))));
}

/* ------------------------------------------------------------------------------------------ */
/* parameter blocks: byte-for-byte the reference's SceneInfo (240 B) and LightInfo (92 B),
 * VulkanRenderer.cs:11-28.  Matrices are 16 floats in memory order; GLSL's M*v on those bytes
 * is out[j] = sum_i v[i]*m[4*i+j]. */

typedef struct {
    float camera_view[16];
    float camera_proj[16];
    float camera_pos[3];    float pad0;
    float dir_light_dir[3]; float pad1;
    float dir_light_color[3]; float pad2;
    float light_space[16];
} orc_scene_info;

typedef struct {
    float model[16];
    float position[3]; float pad0;
    float color[3];
} orc_light_info;

typedef struct {
    int32_t width, height;
    const uint16_t *albedo, *normal, *arm, *position; /* RGBA16F, width*height texels each */
    const uint16_t *background;                       /* RGBA16F or NULL: colour under discarded pixels */
    const orc_scene_info *scene;
    const orc_light_info *lights; int32_t n_lights;
    const float *dir_shadow; int32_t dir_res;         /* D32F dir_res^2 */
    const float *cube_shadow; int32_t cube_res;       /* D32F n_lights * 6 * cube_res^2 */
    const uint16_t *irradiance; int32_t irr_res;      /* RGBA16F 6 * irr_res^2 */
    const uint16_t *prefilter; int32_t pref_res;      /* RGBA16F 6 * pref_res^2 (mip 0) */
    const uint16_t *brdf_lut; int32_t lut_res;        /* RG16F lut_res^2 */
    uint16_t *out_color, *out_threshold;              /* RGBA16F */
    const uint64_t *coverage;                         /* NULL: every light volume covers every pixel;
                                                         else one 64-bit mask per (light, 8x8-pixel tile),
                                                         see orc_light_coverage */
} orc_shade_args;

typedef struct {
    v3 albedo; float alpha; v3 n; v3 pos; float ao, rough, metal;
} gpx;

static inline gpx load_gbuffer(const orc_shade_args *a, size_t p)
{
    gpx g;
    const uint16_t *al = a->albedo + p * 4, *no = a->normal + p * 4, *ar = a->arm + p * 4, *po = a->position + p * 4;
    g.albedo = v3_(h2f(al[0]), h2f(al[1]), h2f(al[2]));
    g.alpha = h2f(al[3]);
    g.n = v3_(h2f(no[0]), h2f(no[1]), h2f(no[2]));
    g.pos = v3_(h2f(po[0]), h2f(po[1]), h2f(po[2]));
    g.ao = h2f(ar[0]); g.rough = h2f(ar[1]); g.metal = h2f(ar[2]);
    return g;
}

/* composition.frag:113-123 / light.frag:85-96 */
static inline float normal_distribution(float NdotH_raw, float rough)
{
    float a = rough * rough;
    float a2 = a * a;
    float NdotH = fmaxf(NdotH_raw, 0.0f);
    float NdotH2 = NdotH * NdotH;
    float denom = fmaf(NdotH2, a2 - 1.0f, 1.0f);
    denom = fmaf(PI_F * denom, denom, 0.0001f);
    return a2 * rcpf_(denom);
}

/* composition.frag:125-130 */
static inline float geometry_schlick(float NdotX, float rough)
{
    float r = rough + 1.0f;
    float k = (r * r) * 0.125f;
    return NdotX * rcpf_(fmaf(NdotX, 1.0f - k, k));
}

/* composition.frag:141-143:  F0 + (1 - F0) * pow(clamp(1 - cosTheta, 0, 1), 5) */
static inline v3 fresnel_schlick(float cosTheta, v3 F0)
{
    float p = pow5(clamp01(1.0f - cosTheta));
    return v3_(fmaf(1.0f - F0.x, p, F0.x), fmaf(1.0f - F0.y, p, F0.y), fmaf(1.0f - F0.z, p, F0.z));
}

static const float LUM_R = 0.2126f, LUM_G = 0.7152f, LUM_B = 0.0722f;
static inline float luminance(v3 c) { return dot3(c, v3_(LUM_R, LUM_G, LUM_B)); }

/* composition.frag:145-168 */
static float dir_shadow(const orc_shade_args *a, v3 pos)
{
    const float *m = a->scene->light_space;
    float p[4];
    for (int j = 0; j < 4; j++)
        p[j] = fmaf(pos.z, m[8 + j], fmaf(pos.y, m[4 + j], fmaf(pos.x, m[j], m[12 + j])));
    float iw = rcpf_(p[3]);
    float px = p[0] * iw, py = p[1] * iw, pz = p[2] * iw;
    float u = fmaf(px, 0.5f, 0.5f), v = fmaf(py, 0.5f, 0.5f);
    float texel = rcpf_((float)a->dir_res);
    float shadow = 0.0f;
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            float d = tex2d_depth_nearest_border(a->dir_shadow, a->dir_res, fmaf((float)x, texel, u), fmaf((float)y, texel, v));
            shadow += pz > d ? 1.0f : 0.0f;
        }
    shadow = shadow * (1.0f / 9.0f);
    if (pz > 1.0f) shadow = 0.0f;
    return shadow;
}

/* light.frag:20-27 */
static const float GRID_DISK[20][3] = {
    {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, 1},
    {1, 1, -1}, {1, -1, -1}, {-1, -1, -1}, {-1, 1, -1},
    {1, 1, 0}, {1, -1, 0}, {-1, -1, 0}, {-1, 1, 0},
    {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1},
    {0, 1, 1}, {0, -1, 1}, {0, -1, -1}, {0, 1, -1}};

/* light.frag:121-141 */
static float point_shadow(const float *cube, int res, v3 pos, v3 lp, v3 cam)
{
    v3 l2f = sub3(pos, lp);
    float cur = sqrtf(dot3(l2f, l2f));
    v3 c2f = sub3(cam, pos);
    float view_distance = sqrtf(dot3(c2f, c2f));
    float disk = fmaf(view_distance, 1.0f / 25.0f, 1.0f) * (1.0f / 25.0f);
    float shadow = 0.0f;
    for (int i = 0; i < 20; i++) {
        v3 d = v3_(fmaf(GRID_DISK[i][0], disk, l2f.x), fmaf(GRID_DISK[i][1], disk, l2f.y), fmaf(GRID_DISK[i][2], disk, l2f.z));
        float closest = texcube_depth_nearest(cube, res, d) * 25.0f;
        if (cur - 0.15f > closest) shadow += 1.0f;
    }
    return shadow * (1.0f / 20.0f);
}

static void shade_pixel(const orc_shade_args *a, size_t p, int first_light, int last_light, int do_composition)
{
    gpx g = load_gbuffer(a, p);
    v3 cam = v3_(a->scene->camera_pos[0], a->scene->camera_pos[1], a->scene->camera_pos[2]);
    uint16_t *oc = a->out_color + p * 4, *ot = a->out_threshold + p * 4;

    /* shared by both shaders: composition.frag:45-54, light.frag:43-51 */
    float om = 1.0f - g.metal;
    v3 F0 = v3_(fmaf(g.albedo.x, g.metal, 0.04f * om), fmaf(g.albedo.y, g.metal, 0.04f * om), fmaf(g.albedo.z, g.metal, 0.04f * om));
    v3 N = normalize3(g.n);
    v3 V = normalize3(sub3(cam, g.pos));
    float NdotV = fmaxf(dot3(N, V), 0.0f);
    int degenerate_normal = !(dot3(g.n, g.n) > 0.0f);

    if (do_composition) {
        /* render-pass clear (0,0,0,1) on both targets (VulkanRenderer.cs:1155-1161 clear values),
         * optionally a background image standing in for the skybox draw on target 0 */
        float c0[3] = {0, 0, 0}, t0[3] = {0, 0, 0};
        if (a->background) { c0[0] = h2f(a->background[p * 4]); c0[1] = h2f(a->background[p * 4 + 1]); c0[2] = h2f(a->background[p * 4 + 2]); }
        if (g.alpha != 0.0f) { /* composition.frag:39-43: discard when alpha == 0 */
            v3 Lo;
            {
                v3 L = normalize3(neg3(v3_(a->scene->dir_light_dir[0], a->scene->dir_light_dir[1], a->scene->dir_light_dir[2])));
                v3 H = normalize3(add3(V, L));
                v3 radiance = v3_(a->scene->dir_light_color[0], a->scene->dir_light_color[1], a->scene->dir_light_color[2]);
                float D = normal_distribution(dot3(N, H), g.rough);
                float NdotL = fmaxf(dot3(N, L), 0.0f);
                float G = geometry_schlick(NdotV, g.rough) * geometry_schlick(NdotL, g.rough);
                v3 F = fresnel_schlick(fmaxf(dot3(H, V), 0.0f), F0);
                v3 kD = v3_((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
                float isd = rcpf_(fmaf(4.0f * NdotV, NdotL, 0.0001f));
                float DG = D * G;
                v3 spec = v3_((DG * F.x) * isd, (DG * F.y) * isd, (DG * F.z) * isd);
                float shadow = dir_shadow(a, g.pos);
                float lit = 1.0f - shadow;
                const float ipi = 1.0f / PI_F;
                Lo.x = ((fmaf(kD.x * g.albedo.x, ipi, spec.x) * radiance.x) * lit) * NdotL;
                Lo.y = ((fmaf(kD.y * g.albedo.y, ipi, spec.y) * radiance.y) * lit) * NdotL;
                Lo.z = ((fmaf(kD.z * g.albedo.z, ipi, spec.z) * radiance.z) * lit) * NdotL;
            }
            {   /* ambient, composition.frag:84-101 */
                v3 F = fresnel_schlick(NdotV, F0);
                v3 kD = v3_((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
                v3 irr = texcube_rgb_linear(a->irradiance, a->irr_res, N);
                /* R = reflect(-V, N) = I - 2*dot(N,I)*N with I = -V */
                v3 I = neg3(V);
                float d2 = 2.0f * dot3(N, I);
                v3 R = v3_(fmaf(-d2, N.x, I.x), fmaf(-d2, N.y, I.y), fmaf(-d2, N.z, I.z));
                /* textureLod(..., roughness*4) resolves to mip 0: MaxLod = 0 on the sampler and a
                 * one-level cube view (VulkanHelper.cs:145,412-413; Cubemap.cs:20) */
                v3 pre = texcube_rgb_linear(a->prefilter, a->pref_res, R);
                float bx, by;
                tex2d_rg_linear_repeat(a->brdf_lut, a->lut_res, a->lut_res, NdotV, g.rough, &bx, &by);
                /* composition.frag:96: prefiltered * (F * brdf.x * brdf.y) */
                v3 spec = v3_(pre.x * ((F.x * bx) * by), pre.y * ((F.y * bx) * by), pre.z * ((F.z * bx) * by));
                Lo.x += fmaf(kD.x, irr.x * g.albedo.x, spec.x) * g.ao;
                Lo.y += fmaf(kD.y, irr.y * g.albedo.y, spec.y) * g.ao;
                Lo.z += fmaf(kD.z, irr.z * g.albedo.z, spec.z) * g.ao;
            }
            c0[0] = Lo.x; c0[1] = Lo.y; c0[2] = Lo.z;
            if (luminance(Lo) > 1.0f) { t0[0] = Lo.x; t0[1] = Lo.y; t0[2] = Lo.z; }
            else { t0[0] = t0[1] = t0[2] = 0.0f; }
        }
        for (int c = 0; c < 3; c++) { oc[c] = f2h(c0[c]); ot[c] = f2h(t0[c]); }
        oc[3] = 0x3C00; ot[3] = 0x3C00;
    }

    /* one additive draw per light, in list order; the attachment is RGBA16F so the running sum
     * is rounded to binary16 after every draw (GraphicsPipelineBuilder.cs:157-187) */
    const int px_x = (int)(p % (size_t)a->width), px_y = (int)(p / (size_t)a->width);
    const int tiles_x = (a->width + 7) / 8, tiles_y = (a->height + 7) / 8;
    for (int li = first_light; li < last_light; li++) {
        const orc_light_info *lt = a->lights + li;
        if (a->coverage) { /* the draw touches only the pixels its light volume covers (light.vert, Light.cs:20-26) */
            uint64_t m = a->coverage[((size_t)li * tiles_y + px_y / 8) * tiles_x + px_x / 8];
            if (!((m >> ((px_y & 7) * 8 + (px_x & 7))) & 1)) continue;
        }
        v3 lp = v3_(lt->position[0], lt->position[1], lt->position[2]);
        v3 out = v3_(0, 0, 0);
        /* A cleared G-buffer texel has normal (0,0,0); normalize() of it is NaN and the shader's
         * result is then implementation-dependent.  With IEEE maxNum semantics every term of the
         * cleared texel collapses to 0, which is what is restated here. */
        if (!degenerate_normal) {
            v3 Lv = sub3(lp, g.pos);
            v3 L = normalize3(Lv);
            v3 H = normalize3(add3(V, L));
            float distance = sqrtf(dot3(Lv, Lv));
            float att = rcpf_(distance * distance);
            v3 radiance = v3_(lt->color[0] * att, lt->color[1] * att, lt->color[2] * att);
            float D = normal_distribution(dot3(N, H), g.rough);
            float NdotL = fmaxf(dot3(N, L), 0.0f);
            float G = geometry_schlick(NdotV, g.rough) * geometry_schlick(NdotL, g.rough);
            v3 F = fresnel_schlick(fmaxf(dot3(H, V), 0.0f), F0);
            v3 kD = v3_(1.0f - F.x, 1.0f - F.y, 1.0f - F.z); /* light.frag:65-67: no (1 - metalness) */
            float isd = rcpf_(fmaf(4.0f * NdotV, NdotL, 0.0001f));
            float DG = D * G;
            v3 spec = v3_((DG * F.x) * isd, (DG * F.y) * isd, (DG * F.z) * isd);
            float shadow = point_shadow(a->cube_shadow + (size_t)li * 6 * a->cube_res * a->cube_res, a->cube_res, g.pos, lp, cam);
            float lit = 1.0f - shadow;
            const float ipi = 1.0f / PI_F;
            out.x = ((fmaf(kD.x * g.albedo.x, ipi, spec.x) * radiance.x) * lit) * NdotL;
            out.y = ((fmaf(kD.y * g.albedo.y, ipi, spec.y) * radiance.y) * lit) * NdotL;
            out.z = ((fmaf(kD.z * g.albedo.z, ipi, spec.z) * radiance.z) * lit) * NdotL;
        }
        int bright = luminance(out) > 1.0f; /* light.frag:78-82: threshold per draw */
        oc[0] = f2h(h2f(oc[0]) + out.x); oc[1] = f2h(h2f(oc[1]) + out.y); oc[2] = f2h(h2f(oc[2]) + out.z);
        ot[0] = f2h(h2f(ot[0]) + (bright ? out.x : 0.0f));
        ot[1] = f2h(h2f(ot[1]) + (bright ? out.y : 0.0f));
        ot[2] = f2h(h2f(ot[2]) + (bright ? out.z : 0.0f));
        oc[3] = 0x3C00; ot[3] = 0x3C00; /* dst.a = src.a = 1 */
    }
}

/* Composition pass (fullscreen composition.frag draw followed by one light.frag draw per light,
 * every light volume covering every pixel) over rows [row_begin, row_end). */
ORC_API void orc_shade(const orc_shade_args *a, int row_begin, int row_end)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = row_begin; y < row_end; y++)
        for (int x = 0; x < a->width; x++)
            shade_pixel(a, (size_t)y * a->width + x, 0, a->n_lights, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* Light-volume coverage.  The reference draws, per light, a sphere mesh scaled to the light's
 * reach and moved to its position (Light.cs:10-26), transformed by light.vert:26
 * (proj * view * model * vec4(pos, 1)), with FRONT faces culled, CCW front, no depth test
 * (GraphicsPipeline.cs:100-117) and the default depth clip 0 <= z <= w.  light.frag then runs on
 * the pixels the remaining (back-facing) triangles cover.  Which pixels those are is decided by
 * the rasteriser of the Vulkan implementation, not by any reference file; restated here from the
 * specification's definition (a pixel belongs to a triangle when its centre lies inside the
 * projected triangle) in homogeneous form, so that triangles crossing the eye plane need no
 * clipping:
 *   with V0,V1,V2 the clip-space (x,y,w) of the corners and p = (xn, yn, 1) the pixel centre in
 *   normalised device coordinates, E0 = p.(V1xV2), E1 = p.(V2xV0), E2 = p.(V0xV1) and
 *   D = V0.(V1xV2).  D > 0 is a back face (the y-flipped projection of Camera.cs:30-35 with
 *   FrontFace.CounterClockwise makes D < 0 front); the centre is inside, in front of the eye,
 *   when E0,E1,E2 >= 0 and E0+E1+E2 > 0; its depth passes the clip when 0 <= sum(Ei*zi) <= D.
 * All of it is evaluated in binary64 from the binary32 clip coordinates with a fixed operation
 * order, edges closed on both sides (two triangles sharing an edge compute exactly negated edge
 * functions, so a centre on the edge belongs to both and the union has no holes).
 * Output: one 64-bit mask per (light, 8x8-pixel tile): bit (y&7)*8 + (x&7). */

typedef struct {
    int32_t width, rows;        /* the image (a strip of the frame) */
    int32_t frame_rows, row0;   /* rows of the whole frame, and the strip's first row in it */
    const orc_scene_info *scene;
    const orc_light_info *lights; int32_t n_lights;
    const float *verts; int32_t n_verts;       /* light-volume mesh, 3 floats per vertex */
    const uint32_t *indices; int32_t n_tris;   /* 3 indices per triangle */
    uint64_t *masks;            /* n_lights * ceil(rows/8) * ceil(width/8) */
} orc_coverage_args;

/* row-major 4x4 product in the row-vector convention: c = a * b, each element one fma chain */
static void mat4_mul(const float *a, const float *b, float *c)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            c[4 * i + j] = fmaf(a[4 * i + 3], b[12 + j], fmaf(a[4 * i + 2], b[8 + j], fmaf(a[4 * i + 1], b[4 + j], a[4 * i] * b[j])));
}

/* light.vert:26 evaluated left to right: (proj * view) * model, then times the vertex */
static void light_mvp(const orc_scene_info *scene, const orc_light_info *lt, float *mvp)
{
    float vp[16];
    mat4_mul(scene->camera_view, scene->camera_proj, vp);
    mat4_mul(lt->model, vp, mvp);
}

static void clip_vertex(const float *mvp, const float *v, float *out4)
{
    for (int j = 0; j < 4; j++)
        out4[j] = fmaf(v[2], mvp[8 + j], fmaf(v[1], mvp[4 + j], v[0] * mvp[j])) + mvp[12 + j];
}

typedef struct { double a[3], b[3], c[3], z[3], D; } cov_tri;

static int cov_setup(const float *c0, const float *c1, const float *c2, cov_tri *t)
{
    const double x0 = c0[0], y0 = c0[1], z0 = c0[2], w0 = c0[3];
    const double x1 = c1[0], y1 = c1[1], z1 = c1[2], w1 = c1[3];
    const double x2 = c2[0], y2 = c2[1], z2 = c2[2], w2 = c2[3];
    t->a[0] = y1 * w2 - w1 * y2; t->b[0] = w1 * x2 - x1 * w2; t->c[0] = x1 * y2 - y1 * x2;
    t->a[1] = y2 * w0 - w2 * y0; t->b[1] = w2 * x0 - x2 * w0; t->c[1] = x2 * y0 - y2 * x0;
    t->a[2] = y0 * w1 - w0 * y1; t->b[2] = w0 * x1 - x0 * w1; t->c[2] = x0 * y1 - y0 * x1;
    t->z[0] = z0; t->z[1] = z1; t->z[2] = z2;
    t->D = fma(w0, t->c[0], fma(y0, t->b[0], x0 * t->a[0]));
    return t->D > 0.0; /* back face; front faces and degenerate triangles are culled */
}

static inline double cov_edge(const cov_tri *t, int i, double xn, double yn) { return fma(t->a[i], xn, fma(t->b[i], yn, t->c[i])); }

static inline int cov_inside(const cov_tri *t, double xn, double yn)
{
    const double e0 = cov_edge(t, 0, xn, yn), e1 = cov_edge(t, 1, xn, yn), e2 = cov_edge(t, 2, xn, yn);
    if (!(e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0 && (e0 + e1) + e2 > 0.0)) return 0;
    const double zs = fma(e2, t->z[2], fma(e1, t->z[1], e0 * t->z[0]));
    return zs >= 0.0 && zs <= t->D;
}

#define COV_MACRO 128 /* triangles are first tested against 128x128-pixel blocks */

ORC_API void orc_light_coverage(const orc_coverage_args *a)
{
    const int tiles_x = (a->width + 7) / 8, tiles_y = (a->rows + 7) / 8;
    const int mx = (a->width + COV_MACRO - 1) / COV_MACRO, my = (a->rows + COV_MACRO - 1) / COV_MACRO;
    const double sx = 2.0 / (double)a->width, sy = 2.0 / (double)a->frame_rows;
    memset(a->masks, 0, (size_t)a->n_lights * tiles_x * tiles_y * sizeof(uint64_t));
    float *clip = (float *)malloc((size_t)a->n_verts * 4 * sizeof(float));
    cov_tri *tris = (cov_tri *)malloc((size_t)a->n_tris * sizeof(cov_tri));
    for (int li = 0; li < a->n_lights; li++) {
        float mvp[16];
        light_mvp(a->scene, a->lights + li, mvp);
        for (int v = 0; v < a->n_verts; v++) clip_vertex(mvp, a->verts + 3 * v, clip + 4 * v);
        int nback = 0;
        for (int t = 0; t < a->n_tris; t++) {
            cov_tri tri;
            if (cov_setup(clip + 4 * a->indices[3 * t], clip + 4 * a->indices[3 * t + 1], clip + 4 * a->indices[3 * t + 2], &tri)) tris[nback++] = tri;
        }
        uint64_t *masks = a->masks + (size_t)li * tiles_x * tiles_y;
#pragma omp parallel for schedule(dynamic, 1)
        for (int mb = 0; mb < mx * my; mb++) {
            const int bx0 = (mb % mx) * COV_MACRO, by0 = (mb / mx) * COV_MACRO;
            const int bx1 = bx0 + COV_MACRO < a->width ? bx0 + COV_MACRO : a->width, by1 = by0 + COV_MACRO < a->rows ? by0 + COV_MACRO : a->rows;
            /* centres of the block's corner pixels */
            const double xa = fma((double)bx0 + 0.5, sx, -1.0), xb = fma((double)(bx1 - 1) + 0.5, sx, -1.0);
            const double ya = fma((double)(a->row0 + by0) + 0.5, sy, -1.0), yb = fma((double)(a->row0 + by1 - 1) + 0.5, sy, -1.0);
            for (int t = 0; t < nback; t++) {
                const cov_tri *tri = tris + t;
                int reject = 0;
                for (int i = 0; i < 3 && !reject; i++) /* an edge with all four corners outside rules the block out */
                    reject = cov_edge(tri, i, xa, ya) < 0.0 && cov_edge(tri, i, xb, ya) < 0.0 && cov_edge(tri, i, xa, yb) < 0.0 && cov_edge(tri, i, xb, yb) < 0.0;
                if (reject) continue;
                /* per 8x8 tile: the same corner test drops the tile, four covered corners fill it
                 * (both follow from the edge functions and the depth being linear over the tile),
                 * only the remaining tiles are tested pixel by pixel */
                for (int ty0 = by0; ty0 < by1; ty0 += 8)
                    for (int tx0 = bx0; tx0 < bx1; tx0 += 8) {
                        const int tx1 = tx0 + 8 < bx1 ? tx0 + 8 : bx1, ty1 = ty0 + 8 < by1 ? ty0 + 8 : by1;
                        uint64_t *m = masks + (size_t)(ty0 / 8) * tiles_x + tx0 / 8;
                        const double cx[2] = {fma((double)tx0 + 0.5, sx, -1.0), fma((double)(tx1 - 1) + 0.5, sx, -1.0)};
                        const double cy[2] = {fma((double)(a->row0 + ty0) + 0.5, sy, -1.0), fma((double)(a->row0 + ty1 - 1) + 0.5, sy, -1.0)};
                        int drop = 0;
                        for (int i = 0; i < 3 && !drop; i++)
                            drop = cov_edge(tri, i, cx[0], cy[0]) < 0.0 && cov_edge(tri, i, cx[1], cy[0]) < 0.0 && cov_edge(tri, i, cx[0], cy[1]) < 0.0 && cov_edge(tri, i, cx[1], cy[1]) < 0.0;
                        if (drop) continue;
                        const int fill = cov_inside(tri, cx[0], cy[0]) && cov_inside(tri, cx[1], cy[0]) && cov_inside(tri, cx[0], cy[1]) && cov_inside(tri, cx[1], cy[1]);
                        for (int y = ty0; y < ty1; y++) {
                            const double yn = fma((double)(a->row0 + y) + 0.5, sy, -1.0);
                            for (int x = tx0; x < tx1; x++) {
                                const uint64_t bit = 1ull << ((y & 7) * 8 + (x & 7));
                                if (*m & bit) continue;
                                if (fill || cov_inside(tri, fma((double)x + 0.5, sx, -1.0), yn)) *m |= bit;
                            }
                        }
                    }
            }
        }
    }
    free(tris);
    free(clip);
}

/* probes for the known-answer tests */
ORC_API void orc_light_clip_vertex(const orc_scene_info *scene, const orc_light_info *lt, const float *v3in, float *out4)
{
    float mvp[16];
    light_mvp(scene, lt, mvp);
    clip_vertex(mvp, v3in, out4);
}

/* ------------------------------------------------------------------------------------------ */
/* Shadow-map generation.  Point lights: VulkanRenderer.cs:325-376 draws every solid model into the
 * six faces of a depth cube per light through point_shadow.vert (world = model * pos),
 * point_shadow.geom (clip = shadowMatrices[face] * world, the world position passed on) and
 * point_shadow.frag (gl_FragDepth = length(world - lightPos) / farPlane), with FRONT faces culled,
 * CCW front, depth test LESS on a cube cleared to 1.0 (GraphicsPipeline.cs:262-264).
 * Directional light: VulkanRenderer.cs:406-428 through depth.vert
 * (clip = lightSpaceMatrix * model * pos), the same raster state, the fixed-function depth.
 * Which texels a triangle covers and how the world position is interpolated belong to the
 * rasteriser; they are restated as in orc_light_coverage (texel centre inside the projected
 * back-facing triangle, homogeneous binary64 edge functions, depth clip 0 <= z <= w), the
 * interpolation being the perspective-correct one: attribute = sum(Ei * ai) / sum(Ei).
 * The result per texel is a minimum over the triangles covering it, so it does not depend on
 * the order triangles are drawn in.
 *
 * Shadow matrices (VulkanRenderer.cs:327-339): CreateLookAt(pos, pos + dir, up) *
 * CreatePerspectiveFieldOfView(pi/2, 1, 1, farPlane) in binary32.  The look-at axes are exact
 * signed unit vectors; 1/tan(pi/4) is taken as exactly 1. */

typedef struct {
    const float *positions; int32_t n_verts;   /* 3 floats per vertex */
    const uint32_t *indices; int32_t n_tris;   /* 3 per triangle */
    float model[16];                           /* DrawCall.ModelMatrix, row-major, row-vector convention */
} orc_mesh_draw;

typedef struct {
    const orc_mesh_draw *draws; int32_t n_draws;
    const orc_light_info *lights; int32_t n_lights;   /* positions are read */
    int32_t res; float far_plane;
    float *out;                                        /* n_lights * 6 * res * res, row-major faces */
} orc_point_shadow_args;

typedef struct {
    const orc_mesh_draw *draws; int32_t n_draws;
    const float *light_space;                          /* SceneInfo.lightSpaceMatrix */
    int32_t res;
    float *out;                                        /* res * res */
} orc_dir_shadow_args;

static const float SHADOW_DIR[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
static const float SHADOW_UP[6][3] = {{0, -1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, -1, 0}, {0, -1, 0}};

/* CreateLookAt(eye, eye + dir, up) * CreatePerspectiveFieldOfView(pi/2, 1, near = 1, far) */
static void shadow_face_matrix(const float *eye, int face, float far_plane, float *out16)
{
    const float *d = SHADOW_DIR[face], *u = SHADOW_UP[face];
    const float z[3] = {-d[0], -d[1], -d[2]};                              /* normalize(eye - target) */
    const float x[3] = {u[1] * z[2] - u[2] * z[1], u[2] * z[0] - u[0] * z[2], u[0] * z[1] - u[1] * z[0]}; /* cross(up, z), already unit */
    const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]}; /* cross(z, x) */
    float view[16] = {x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0, 0, 0, 0, 1};
    view[12] = -fmaf(x[2], eye[2], fmaf(x[1], eye[1], x[0] * eye[0]));
    view[13] = -fmaf(y[2], eye[2], fmaf(y[1], eye[1], y[0] * eye[0]));
    view[14] = -fmaf(z[2], eye[2], fmaf(z[1], eye[1], z[0] * eye[0]));
    const float range = far_plane * rcpf_(1.0f - far_plane);              /* far / (near - far), near = 1 */
    const float proj[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, range, -1, 0, 0, range, 0};
    mat4_mul(view, proj, out16);
}

/* GLSL mat4 * vec4 on row-major bytes: out[j] = sum_i v[i] * m[4i + j] */
static void mat4_vec4(const float *m, const float *v, float *out4)
{
    for (int j = 0; j < 4; j++)
        out4[j] = fmaf(v[3], m[12 + j], fmaf(v[2], m[8 + j], fmaf(v[1], m[4 + j], v[0] * m[j])));
}

/* Conservative texel bounding box of the part of a clip-space triangle with z >= 0 (and w > 0).
 * Returns 0 when nothing can be covered. */
static int shadow_bbox(const float (*c)[4], int res, int *bx0, int *by0, int *bx1, int *by1)
{
    double poly[8][4];
    int n = 0;
    for (int i = 0; i < 3; i++) {
        const float *a = c[i], *b = c[(i + 1) % 3];
        const int ain = a[2] >= 0.0f && a[3] > 0.0f, bin = b[2] >= 0.0f && b[3] > 0.0f;
        if (ain) { for (int k = 0; k < 4; k++) poly[n][k] = a[k]; n++; }
        if (ain != bin) {
            /* crossing of z = 0 (or of w = 0 when z stays positive): take both candidates' safer one */
            double ta = (double)a[2], tb = (double)b[2];
            double t = (ta != tb) ? ta / (ta - tb) : 0.5;
            if (!(t >= 0.0 && t <= 1.0)) { double wa = a[3], wb = b[3]; t = (wa != wb) ? wa / (wa - wb) : 0.5; }
            for (int k = 0; k < 4; k++) poly[n][k] = (double)a[k] + t * ((double)b[k] - (double)a[k]);
            n++;
        }
    }
    if (n == 0) return 0;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int i = 0; i < n; i++) {
        const double w = poly[i][3];
        if (!(w > 1e-12)) { xmin = ymin = -1e300; xmax = ymax = 1e300; break; }  /* cannot bound: whole face */
        const double x = poly[i][0] / w, y = poly[i][1] / w;
        if (x < xmin) xmin = x;
        if (x > xmax) xmax = x;
        if (y < ymin) ymin = y;
        if (y > ymax) ymax = y;
    }
    const double half = 0.5 * (double)res;
    double fx0 = floor((xmin + 1.0) * half - 0.5) - 1.0, fx1 = ceil((xmax + 1.0) * half - 0.5) + 1.0;
    double fy0 = floor((ymin + 1.0) * half - 0.5) - 1.0, fy1 = ceil((ymax + 1.0) * half - 0.5) + 1.0;
    if (!(fx0 > 0.0)) fx0 = 0.0;
    if (!(fy0 > 0.0)) fy0 = 0.0;
    if (!(fx1 < (double)(res - 1))) fx1 = (double)(res - 1);
    if (!(fy1 < (double)(res - 1))) fy1 = (double)(res - 1);
    if (fx0 > fx1 || fy0 > fy1) return 0;
    *bx0 = (int)fx0; *by0 = (int)fy0; *bx1 = (int)fx1; *by1 = (int)fy1;
    return 1;
}

/* Rasterises one triangle into one res x res depth image.  clip: the three clip-space corners;
 * world: their world positions (point lights) or NULL (directional: fixed-function depth). */
/* point_shadow.frag:11-16 */
static float point_shadow_depth(const float *p, const float *light_pos, float inv_far)
{
    v3 d = v3_(p[0] - light_pos[0], p[1] - light_pos[1], p[2] - light_pos[2]);
    return sqrtf(dot3(d, d)) * inv_far;
}

/* point_shadow.vert:12 (world = model * pos) and point_shadow.geom:19-20 (clip = shadowMatrices[face] * world) */
static void point_shadow_vertex(const float *model, const float *face_matrix, const float *pos3, float *world4, float *clip4)
{
    const float v4[4] = {pos3[0], pos3[1], pos3[2], 1.0f};
    mat4_vec4(model, v4, world4);
    mat4_vec4(face_matrix, world4, clip4);
}

static void shadow_raster(const float (*clip)[4], const float (*world)[4], const float *light_pos, float inv_far, int res, float *img)
{
    cov_tri t;
    if (!cov_setup(clip[0], clip[1], clip[2], &t)) return;
    int bx0, by0, bx1, by1;
    if (!shadow_bbox(clip, res, &bx0, &by0, &bx1, &by1)) return;
    const double sc = 2.0 / (double)res;
    const double inv_D = 1.0 / t.D;  /* one division per triangle */
    for (int j = by0; j <= by1; j++) {
        const double yn = fma((double)j + 0.5, sc, -1.0);
        for (int i = bx0; i <= bx1; i++) {
            const double xn = fma((double)i + 0.5, sc, -1.0);
            const double e0 = cov_edge(&t, 0, xn, yn), e1 = cov_edge(&t, 1, xn, yn), e2 = cov_edge(&t, 2, xn, yn);
            const double s = (e0 + e1) + e2;
            if (!(e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0 && s > 0.0)) continue;
            const double zs = fma(e2, t.z[2], fma(e1, t.z[1], e0 * t.z[0]));
            if (!(zs >= 0.0 && zs <= t.D)) continue;
            float depth;
            if (world) { /* point_shadow.frag:11-16 on the interpolated world position */
                /* Coverage is decided in binary64 (above); the attribute is interpolated the way graphics hardware does it,
                 * in binary32: the edge values rounded to binary32 are the perspective-correct weights, sum(Ei * Pi) * (1 / sum(Ei)). */
                float p[3];
                const float f0 = (float)e0, f1 = (float)e1, f2 = (float)e2;
                const float inv_s = rcpf_((f0 + f1) + f2);
                for (int k = 0; k < 3; k++)
                    p[k] = fmaf(f2, world[2][k], fmaf(f1, world[1][k], f0 * world[0][k])) * inv_s;
                depth = point_shadow_depth(p, light_pos, inv_far);
            } else {
                depth = (float)(zs * inv_D) + 0.0f; /* z / w of the fixed-function pipeline as z * (1/w) (-0 becomes +0) */
            }
            float *o = img + (size_t)j * res + i;
            if (depth < *o) *o = depth; /* CompareOp.Less, depth writes on */
        }
    }
}

ORC_API void orc_point_shadow_maps(const orc_point_shadow_args *a)
{
    const size_t face_texels = (size_t)a->res * a->res;
    const float inv_far = rcpf_(a->far_plane);
    for (size_t i = 0; i < (size_t)a->n_lights * 6 * face_texels; i++) a->out[i] = 1.0f; /* clear value */
#pragma omp parallel for schedule(dynamic, 1)
    for (int lf = 0; lf < a->n_lights * 6; lf++) {
        const int li = lf / 6, face = lf % 6;
        const float *lp = a->lights[li].position;
        float m[16];
        shadow_face_matrix(lp, face, a->far_plane, m);
        float *img = a->out + (size_t)lf * face_texels;
        for (int d = 0; d < a->n_draws; d++) {
            const orc_mesh_draw *dr = a->draws + d;
            for (int t = 0; t < dr->n_tris; t++) {
                float world[3][4], clip[3][4];
                for (int k = 0; k < 3; k++) {
                    point_shadow_vertex(dr->model, m, dr->positions + 3 * (size_t)dr->indices[3 * t + k], world[k], clip[k]);
                }
                shadow_raster(clip, world, lp, inv_far, a->res, img);
            }
        }
    }
}

ORC_API void orc_dir_shadow_map(const orc_dir_shadow_args *a)
{
    const size_t texels = (size_t)a->res * a->res;
    for (size_t i = 0; i < texels; i++) a->out[i] = 1.0f;
    for (int d = 0; d < a->n_draws; d++) {
        const orc_mesh_draw *dr = a->draws + d;
        float mvp[16];
        mat4_mul(dr->model, a->light_space, mvp);         /* depth.vert:13, associated as lightSpace * model */
        for (int t = 0; t < dr->n_tris; t++) {
            float clip[3][4];
            for (int k = 0; k < 3; k++) {
                const float *v = dr->positions + 3 * (size_t)dr->indices[3 * t + k];
                const float v4[4] = {v[0], v[1], v[2], 1.0f};
                mat4_vec4(mvp, v4, clip[k]);
            }
            shadow_raster(clip, NULL, NULL, 0.0f, a->res, a->out);
        }
    }
}

ORC_API void orc_shadow_face_matrix(const float *eye, int face, float far_plane, float *out16) { shadow_face_matrix(eye, face, far_plane, out16); }

/* Single invocations of the shadow passes' shaders, for the tests that pin them to the reference's binaries. */
ORC_API void orc_point_shadow_vertex(const float *model, const float *face_matrix, const float *pos3, float *world4, float *clip4)
{
    point_shadow_vertex(model, face_matrix, pos3, world4, clip4);
}

ORC_API float orc_point_shadow_depth(const float *world3, const float *light_pos, float far_plane)
{
    return point_shadow_depth(world3, light_pos, rcpf_(far_plane));
}

ORC_API void orc_depth_vertex(const float *model, const float *light_space, const float *pos3, float *clip4)
{
    float mvp[16];
    const float v4[4] = {pos3[0], pos3[1], pos3[2], 1.0f};
    mat4_mul(model, light_space, mvp);  /* depth.vert:13, as orc_dir_shadow_map */
    mat4_vec4(mvp, v4, clip4);
}

/* ------------------------------------------------------------------------------------------ */
/* bloom_blur.frag:14-33.  wrap = REPEAT for pass 1 (reads the thresholded target through
 * textureSampler), CLAMP for pass 2 (bloomSampler): VulkanRenderer.cs:231-235.
 * true_bilinear = 0: fetch the texel whose centre the coordinate names;
 * true_bilinear = 1: evaluate the LINEAR filter with the shader's own fp32 coordinates
 * (uv = (x+0.5)/W, offset = i * (1/W)), which shows how little weight leaks to neighbours. */

static const float BLOOM_W[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};

static inline void fetch_rgb(const uint16_t *img, int w, int x, int y, float *o)
{
    const uint16_t *p = img + ((size_t)y * w + x) * 4;
    o[0] = h2f(p[0]); o[1] = h2f(p[1]); o[2] = h2f(p[2]);
}

static void sample_rgb_linear(const uint16_t *img, int w, int h, int wrap, float s, float t, float *o)
{
    float x = s * (float)w - 0.5f, y = t * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float al = x - fx, be = y - fy;
    int i0 = wrap_i((int)fx, w, wrap), i1 = wrap_i((int)fx + 1, w, wrap);
    int j0 = wrap_i((int)fy, h, wrap), j1 = wrap_i((int)fy + 1, h, wrap);
    float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
    float a[3], b[3], c[3], d[3];
    fetch_rgb(img, w, i0, j0, a); fetch_rgb(img, w, i1, j0, b); fetch_rgb(img, w, i0, j1, c); fetch_rgb(img, w, i1, j1, d);
    for (int k = 0; k < 3; k++) o[k] = fmaf(w11, d[k], fmaf(w01, c[k], fmaf(w10, b[k], w00 * a[k])));
}

ORC_API void orc_bloom_blur(const uint16_t *src, uint16_t *dst, int w, int h, int horizontal, int wrap, int true_bilinear)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc[3], t[3];
            float u = ((float)x + 0.5f) * rcpf_((float)w), v = ((float)y + 0.5f) * rcpf_((float)h);
            float ox = rcpf_((float)w), oy = rcpf_((float)h);
            if (true_bilinear) sample_rgb_linear(src, w, h, wrap, u, v, t); else fetch_rgb(src, w, x, y, t);
            for (int c = 0; c < 3; c++) acc[c] = t[c] * BLOOM_W[0];
            for (int i = 1; i < 5; i++) {
                for (int sgn = 1; sgn >= -1; sgn -= 2) { /* +offset tap first, then -offset */
                    if (true_bilinear) {
                        float du = horizontal ? ox * (float)i : 0.0f, dv = horizontal ? 0.0f : oy * (float)i;
                        sample_rgb_linear(src, w, h, wrap, sgn > 0 ? u + du : u - du, sgn > 0 ? v + dv : v - dv, t);
                    } else {
                        int xx = horizontal ? wrap_i(x + sgn * i, w, wrap) : x;
                        int yy = horizontal ? y : wrap_i(y + sgn * i, h, wrap);
                        fetch_rgb(src, w, xx, yy, t);
                    }
                    for (int c = 0; c < 3; c++) acc[c] = fmaf(t[c], BLOOM_W[i], acc[c]);
                }
            }
            uint16_t *o = dst + ((size_t)y * w + x) * 4;
            o[0] = f2h(acc[0]); o[1] = f2h(acc[1]); o[2] = f2h(acc[2]); o[3] = 0x3C00;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* postprocess.frag:11-38.  The 8-bit target stores round(clamp(x,0,1)*255).  srgb_target = 1
 * additionally applies the sRGB transfer function an _SRGB swapchain format would apply on store
 * (SCDevice.cs:655-666 prefers R8G8B8A8_SRGB when the surface offers it). */

static inline float aces(float c)
{
    float num = c * fmaf(2.51f, c, 0.03f);
    float den = fmaf(c, fmaf(2.43f, c, 0.59f), 0.14f);
    return clamp01(num * rcpf_(den));
}

static inline float linear_to_srgb(float x)
{
    return x <= 0.0031308f ? 12.92f * x : fmaf(1.055f, powf(x, 1.0f / 2.4f), -0.055f);
}

ORC_API void orc_postprocess(const uint16_t *hdr, const uint16_t *bloom, uint8_t *out, int w, int h, int srgb_target)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t p = (size_t)y * w + x;
            for (int c = 0; c < 3; c++) {
                float v = h2f(hdr[p * 4 + c]) + h2f(bloom[p * 4 + c]);
                float m = aces(v);
                float gam = powf(m, 1.0f / 2.2f);
                if (srgb_target) gam = linear_to_srgb(clamp01(gam));
                out[p * 4 + c] = (uint8_t)floorf(fmaf(clamp01(gam), 255.0f, 0.5f));
            }
            out[p * 4 + 3] = 255;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* IBL precompute                                                                               */

/* Direction through the centre of texel (i,j) of face f of an n^2 cube.  Follows from the six
 * LookAt matrices of VulkanRenderer.cs:656-665 with a 90 degree projection and Vulkan's y-down
 * framebuffer; it is the inverse of cube_face() above. */
static inline v3 face_dir(int f, int i, int j, int n)
{
    float rn = rcpf_((float)n);
    float s = fmaf((float)(2 * i + 1), rn, -1.0f), t = fmaf((float)(2 * j + 1), rn, -1.0f);
    switch (f) {
    case 0: return v3_(1.0f, -t, -s);
    case 1: return v3_(-1.0f, -t, s);
    case 2: return v3_(s, 1.0f, t);
    case 3: return v3_(s, -1.0f, -t);
    case 4: return v3_(s, -t, 1.0f);
    default: return v3_(-s, -t, -1.0f);
    }
}

ORC_API void orc_face_dir(int f, int i, int j, int n, float *out3)
{
    v3 d = face_dir(f, i, j, n);
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

enum { ORC_SRC_RGBA8_SRGB = 0, ORC_SRC_RGBA16F = 1 };

static float SRGB_LUT[256];
static int srgb_lut_ready = 0;
static void build_srgb_lut(void)
{
    if (srgb_lut_ready) return;
    for (int i = 0; i < 256; i++) {
        float c = (float)i * (1.0f / 255.0f);
        SRGB_LUT[i] = c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f);
    }
    srgb_lut_ready = 1;
}

static inline void fetch_src(const void *src, int fmt, int w, int x, int y, float *o)
{
    if (fmt == ORC_SRC_RGBA8_SRGB) {
        const uint8_t *p = (const uint8_t *)src + ((size_t)y * w + x) * 4;
        o[0] = SRGB_LUT[p[0]]; o[1] = SRGB_LUT[p[1]]; o[2] = SRGB_LUT[p[2]];
    } else {
        fetch_rgb((const uint16_t *)src, w, x, y, o);
    }
}

/* equirectangular_to_cubemap.frag:8-24; source sampled LINEAR / REPEAT (textureSampler,
 * VulkanRenderer.cs:683); an RGBA8 source is an _SRGB image (Texture.cs:21). */
ORC_API void orc_equirect_to_cube(const void *src, int src_w, int src_h, int src_fmt, uint16_t *cube, int n)
{
    build_srgb_lut();
#pragma omp parallel for schedule(static) collapse(2)
    for (int f = 0; f < 6; f++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                v3 d = normalize3(face_dir(f, i, j, n));
                float u = fmaf(atan2f(d.z, d.x), 0.1591f, 0.5f);
                float v = 1.0f - fmaf(asinf(d.y), 0.3183f, 0.5f);
                float x = u * (float)src_w - 0.5f, y = v * (float)src_h - 0.5f;
                float fx = floorf(x), fy = floorf(y);
                float al = x - fx, be = y - fy;
                int i0 = wrap_i((int)fx, src_w, ORC_WRAP_REPEAT), i1 = wrap_i((int)fx + 1, src_w, ORC_WRAP_REPEAT);
                int j0 = wrap_i((int)fy, src_h, ORC_WRAP_REPEAT), j1 = wrap_i((int)fy + 1, src_h, ORC_WRAP_REPEAT);
                float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
                float a[3], b[3], c[3], e[3];
                fetch_src(src, src_fmt, src_w, i0, j0, a); fetch_src(src, src_fmt, src_w, i1, j0, b);
                fetch_src(src, src_fmt, src_w, i0, j1, c); fetch_src(src, src_fmt, src_w, i1, j1, e);
                uint16_t *o = cube + (((size_t)f * n + j) * n + i) * 4;
                for (int k = 0; k < 3; k++)
                    o[k] = f2h(fmaf(w11, e[k], fmaf(w01, c[k], fmaf(w10, b[k], w00 * a[k]))));
                o[3] = 0x3C00;
            }
}

/* The region of the environment cube a LINEAR lookup along r starts in (order = 2 of orc_irradiance): every
 * face is cut into regions of 512 x 512 texels; a lookup belongs to the region that holds the base texel of
 * its 2x2 footprint, clamped to the face; regions are numbered (row * per_side + column) * 6 + face. */
static inline int cube_lookup_region(int n, v3 r)
{
    int face; float s, t;
    cube_st(r, &face, &s, &t);
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    int i0 = f2i_sat(floorf(x)), j0 = f2i_sat(floorf(y));
    i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);
    j0 = j0 < 0 ? 0 : (j0 > n - 1 ? n - 1 : j0);
    int per_side = (n + 511) >> 9;
    return ((j0 >> 9) * per_side + (i0 >> 9)) * 6 + face;
}

/* Order = 3 of orc_irradiance cuts the faces differently.  The base texel (i0, j0) of a lookup's 2x2 footprint lies in
 * [-1, n-1]^2 (-1: the footprint reaches over the face's edge); shifted by one, (I0, J0) = (i0 + 1, j0 + 1) lies in
 * [0, n]^2.  Cells of 3 x 3 such positions (n/3 + 1 per side) are grouped into square regions of `cells_per_region`
 * cells per side, chosen as the most even split that keeps a region at or below 128 cells (384 texels) per side;
 * regions are numbered (row * per_side + column) * 6 + face. */
static inline void cube_region_grid3(int n, int *per_side, int *cells_per_region)
{
    int cells = n / 3 + 1;
    *per_side = (cells + 127) / 128;
    *cells_per_region = (cells + *per_side - 1) / *per_side;
}

static inline int cube_lookup_region3(int n, v3 r)
{
    int face; float s, t;
    cube_st(r, &face, &s, &t);
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    int I0 = f2i_sat(floorf(x)) + 1, J0 = f2i_sat(floorf(y)) + 1;
    I0 = I0 < 0 ? 0 : (I0 > n ? n : I0);   /* only non-finite directions ever leave [0, n] */
    J0 = J0 < 0 ? 0 : (J0 > n ? n : J0);
    int per_side, cpr;
    cube_region_grid3(n, &per_side, &cpr);
    return (((J0 / 3) / cpr) * per_side + (I0 / 3) / cpr) * 6 + face;
}

ORC_API int orc_cube_lookup_region3(int n, const float *dir3)
{
    return cube_lookup_region3(n, v3_(dir3[0], dir3[1], dir3[2]));
}

/* irradiance.frag:10-39.  The loop variables are binary32 running sums, which gives 252 phi
 * values and 63 theta values.  order = 0: one running sum over all samples, phi outer / theta
 * inner, as the shader does.  order = 1: each phi slice is summed on its own (theta ascending,
 * starting from 0) and the slice sums are then added in phi order -- the order a kernel that
 * spreads phi slices over lanes can reproduce bit for bit.  order = 2: the samples are grouped by
 * the region of the environment their lookup starts in (cube_lookup_region); within a region the
 * k-th sample (phi outer / theta inner order) goes to accumulator k mod 64, the 64 accumulators are
 * combined pairwise (partner = index xor 32, 16, 8, 4, 2, 1), and the region sums are added in
 * region order -- the order a kernel that works through the environment region by region, so that
 * its lookups hit the cache, can reproduce bit for bit.  order = 3: the samples are grouped by the region of
 * cube_lookup_region3; within a (texel, region) pair every phi row is summed on its own (theta ascending, starting
 * from 0, only the samples that belong to the region), the row sums are added in phi order, and the region sums in
 * region order -- what a kernel that gives every lane a phi row of a pair reproduces bit for bit. */
ORC_API void orc_irradiance(const uint16_t *env, int env_n, uint16_t *out, int n, int order)
{
    float phis[512], thetas[128];
    int nphi = 0, ntheta = 0;
    for (float phi = 0.0f; phi < 2.0f * PI_F; phi += 0.025f) phis[nphi++] = phi;
    for (float th = 0.0f; th < 0.5f * PI_F; th += 0.025f) thetas[ntheta++] = th;
    float inv_count = rcpf_((float)(nphi * ntheta));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int f = 0; f < 6; f++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                v3 N = normalize3(face_dir(f, i, j, n));
                /* right = normalize(cross((0,1,0), N)) = normalize((N.z, 0, -N.x)) */
                v3 right = normalize3(v3_(N.z, 0.0f, -N.x));
                v3 up = normalize3(cross3(N, right));
                float irr[3] = {0, 0, 0};
                if (order == 2) {
                    const int per_side = (env_n + 511) >> 9, n_regions = 6 * per_side * per_side;
                    float (*lanes)[64][3] = calloc((size_t)n_regions, sizeof(*lanes));
                    int *count = calloc((size_t)n_regions, sizeof(int));
                    for (int a = 0; a < nphi; a++) {
                        float cp = cosf(phis[a]), sp = sinf(phis[a]);
                        for (int b = 0; b < ntheta; b++) {
                            float st = sinf(thetas[b]), ct = cosf(thetas[b]);
                            float tx = st * cp, ty = st * sp, tz = ct;
                            v3 sv = v3_(fmaf(tz, N.x, fmaf(ty, up.x, tx * right.x)),
                                        fmaf(tz, N.y, fmaf(ty, up.y, tx * right.y)),
                                        fmaf(tz, N.z, fmaf(ty, up.z, tx * right.z)));
                            v3 c = texcube_rgb_linear(env, env_n, sv);
                            int r = cube_lookup_region(env_n, sv);
                            float *acc = lanes[r][count[r]++ & 63];
                            acc[0] = fmaf(c.x * ct, st, acc[0]);
                            acc[1] = fmaf(c.y * ct, st, acc[1]);
                            acc[2] = fmaf(c.z * ct, st, acc[2]);
                        }
                    }
                    for (int r = 0; r < n_regions; r++) {
                        for (int d = 32; d >= 1; d >>= 1)
                            for (int l = 0; l < d; l++)   /* every lane adds its partner: lanes l and l ^ d end up equal */
                                for (int k = 0; k < 3; k++) lanes[r][l][k] = lanes[r][l][k] + lanes[r][l ^ d][k];
                        for (int k = 0; k < 3; k++) irr[k] = irr[k] + lanes[r][0][k];
                    }
                    free(lanes); free(count);
                }
                if (order == 3) {
                    int per_side, cpr;
                    cube_region_grid3(env_n, &per_side, &cpr);
                    const int n_regions = 6 * per_side * per_side;
                    float (*pair)[3] = calloc((size_t)n_regions, sizeof(*pair));   /* sum of the rows added so far */
                    float (*row)[3] = calloc((size_t)n_regions, sizeof(*row));     /* the current phi row's sum */
                    unsigned char *touched = calloc((size_t)n_regions, 1);
                    int *list = malloc((size_t)n_regions * sizeof(int));
                    for (int a = 0; a < nphi; a++) {
                        float cp = cosf(phis[a]), sp = sinf(phis[a]);
                        int n_list = 0;
                        for (int b = 0; b < ntheta; b++) {
                            float st = sinf(thetas[b]), ct = cosf(thetas[b]);
                            float tx = st * cp, ty = st * sp, tz = ct;
                            v3 sv = v3_(fmaf(tz, N.x, fmaf(ty, up.x, tx * right.x)),
                                        fmaf(tz, N.y, fmaf(ty, up.y, tx * right.y)),
                                        fmaf(tz, N.z, fmaf(ty, up.z, tx * right.z)));
                            v3 c = texcube_rgb_linear(env, env_n, sv);
                            int r = cube_lookup_region3(env_n, sv);
                            if (!touched[r]) { touched[r] = 1; list[n_list++] = r; row[r][0] = row[r][1] = row[r][2] = 0.0f; }
                            row[r][0] = fmaf(c.x * ct, st, row[r][0]);
                            row[r][1] = fmaf(c.y * ct, st, row[r][1]);
                            row[r][2] = fmaf(c.z * ct, st, row[r][2]);
                        }
                        for (int q = 0; q < n_list; q++) {
                            int r = list[q];
                            for (int k = 0; k < 3; k++) pair[r][k] = pair[r][k] + row[r][k];
                            touched[r] = 0;
                        }
                    }
                    for (int r = 0; r < n_regions; r++)
                        for (int k = 0; k < 3; k++) irr[k] = irr[k] + pair[r][k];
                    free(pair); free(row); free(touched); free(list);
                }
                for (int a = 0; a < nphi && order < 2; a++) {
                    float cp = cosf(phis[a]), sp = sinf(phis[a]);
                    float part[3] = {0, 0, 0};
                    float *acc = order ? part : irr;
                    for (int b = 0; b < ntheta; b++) {
                        float st = sinf(thetas[b]), ct = cosf(thetas[b]);
                        float tx = st * cp, ty = st * sp, tz = ct;
                        v3 sv = v3_(fmaf(tz, N.x, fmaf(ty, up.x, tx * right.x)),
                                    fmaf(tz, N.y, fmaf(ty, up.y, tx * right.y)),
                                    fmaf(tz, N.z, fmaf(ty, up.z, tx * right.z)));
                        v3 c = texcube_rgb_linear(env, env_n, sv);
                        acc[0] = fmaf(c.x * ct, st, acc[0]);
                        acc[1] = fmaf(c.y * ct, st, acc[1]);
                        acc[2] = fmaf(c.z * ct, st, acc[2]);
                    }
                    if (order) { irr[0] += part[0]; irr[1] += part[1]; irr[2] += part[2]; }
                }
                uint16_t *o = out + (((size_t)f * n + j) * n + i) * 4;
                for (int k = 0; k < 3; k++) o[k] = f2h((PI_F * irr[k]) * inv_count);
                o[3] = 0x3C00;
            }
}

/* prefilter.frag:65-73 */
static inline float radical_inverse(uint32_t bits)
{
    bits = (bits << 16) | (bits >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    return (float)bits * 2.3283064365386963e-10f;
}

ORC_API float orc_radical_inverse(uint32_t bits) { return radical_inverse(bits); }

/* prefilter.frag:80-101 / brdf.frag:37-58 */
static inline v3 importance_sample_ggx(float xi_x, float xi_y, v3 N, float rough)
{
    float a = rough * rough;
    float phi = (2.0f * PI_F) * xi_x;
    float cos_t = sqrtf((1.0f - xi_y) * rcpf_(fmaf(a * a - 1.0f, xi_y, 1.0f)));
    float sin_t = sqrtf(fmaf(-cos_t, cos_t, 1.0f));
    float hx = cosf(phi) * sin_t, hy = sinf(phi) * sin_t, hz = cos_t;
    v3 tangent = fabsf(N.z) < 0.999f ? normalize3(v3_(-N.y, N.x, 0.0f))   /* cross((0,0,1), N) */
                                     : normalize3(v3_(0.0f, -N.z, N.y));  /* cross((1,0,0), N) */
    v3 bitangent = cross3(N, tangent);
    v3 sv = v3_(fmaf(N.x, hz, fmaf(bitangent.x, hy, tangent.x * hx)),
                fmaf(N.y, hz, fmaf(bitangent.y, hy, tangent.y * hx)),
                fmaf(N.z, hz, fmaf(bitangent.z, hy, tangent.z * hx)));
    return normalize3(sv);
}

/* prefilter.frag:21-63 for one mip: n^2 texels per face at the given roughness.  The mip level
 * the shader computes for its textureLod is not restated: the sampler clamps every lod to 0. */
ORC_API void orc_prefilter(const uint16_t *env, int env_n, uint16_t *out, int n, float roughness)
{
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int f = 0; f < 6; f++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                v3 N = normalize3(face_dir(f, i, j, n));
                v3 V = N;
                float col[3] = {0, 0, 0}, total = 0.0f;
                for (uint32_t s = 0; s < 1024u; s++) {
                    v3 H = importance_sample_ggx((float)s * (1.0f / 1024.0f), radical_inverse(s), N, roughness);
                    float vh2 = 2.0f * dot3(V, H);
                    v3 L = normalize3(v3_(fmaf(vh2, H.x, -V.x), fmaf(vh2, H.y, -V.y), fmaf(vh2, H.z, -V.z)));
                    float NdotL = fmaxf(dot3(N, L), 0.0f);
                    if (NdotL > 0.0f) {
                        v3 c = texcube_rgb_linear(env, env_n, L);
                        col[0] = fmaf(c.x, NdotL, col[0]); col[1] = fmaf(c.y, NdotL, col[1]); col[2] = fmaf(c.z, NdotL, col[2]);
                        total += NdotL;
                    }
                }
                float iw = rcpf_(total);
                uint16_t *o = out + (((size_t)f * n + j) * n + i) * 4;
                for (int k = 0; k < 3; k++) o[k] = f2h(col[k] * iw);
                o[3] = 0x3C00;
            }
}

/* brdf.frag:60-69 */
static inline float geometry_schlick_ibl(float NdotX, float rough)
{
    float k = (rough * rough) * 0.5f;
    return NdotX * rcpf_(fmaf(NdotX, 1.0f - k, k));
}

/* brdf.frag:81-119 over an n^2 RG16F image; texel (i,j): NdotV = (i+0.5)/n, roughness = (j+0.5)/n */
ORC_API void orc_brdf_lut(uint16_t *out, int n)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            float NdotV = ((float)i + 0.5f) * rcpf_((float)n), rough = ((float)j + 0.5f) * rcpf_((float)n);
            v3 V = v3_(sqrtf(fmaf(-NdotV, NdotV, 1.0f)), 0.0f, NdotV);
            v3 N = v3_(0.0f, 0.0f, 1.0f);
            float A = 0.0f, B = 0.0f;
            for (uint32_t s = 0; s < 1024u; s++) {
                v3 H = importance_sample_ggx((float)s * (1.0f / 1024.0f), radical_inverse(s), N, rough);
                float vh = dot3(V, H);
                float vh2 = 2.0f * vh;
                v3 L = normalize3(v3_(fmaf(vh2, H.x, -V.x), fmaf(vh2, H.y, -V.y), fmaf(vh2, H.z, -V.z)));
                float NdotL = fmaxf(L.z, 0.0f), NdotH = fmaxf(H.z, 0.0f), VdotH = fmaxf(vh, 0.0f);
                if (NdotL > 0.0f) {
                    float G = geometry_schlick_ibl(NdotL, rough) * geometry_schlick_ibl(fmaxf(dot3(N, V), 0.0f), rough);
                    float G_vis = (G * VdotH) * rcpf_(NdotH * NdotV);
                    float Fc = pow5(1.0f - VdotH);
                    A = fmaf(1.0f - Fc, G_vis, A);
                    B = fmaf(Fc, G_vis, B);
                }
            }
            out[((size_t)j * n + i) * 2] = f2h(A * (1.0f / 1024.0f));
            out[((size_t)j * n + i) * 2 + 1] = f2h(B * (1.0f / 1024.0f));
        }
}

/* ------------------------------------------------------------------------------------------ */
/* scalar probes used by the known-answer tests                                                 */

ORC_API float orc_aces(float c) { return aces(c); }
ORC_API float orc_normal_distribution(float ndoth, float rough) { return normal_distribution(ndoth, rough); }
ORC_API float orc_geometry_schlick(float ndotx, float rough) { return geometry_schlick(ndotx, rough); }
ORC_API void orc_texcube_rgb_linear(const uint16_t *cube, int n, const float *dir, float *out3)
{
    v3 c = texcube_rgb_linear(cube, n, v3_(dir[0], dir[1], dir[2]));
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
ORC_API float orc_texcube_depth_nearest(const float *cube, int n, const float *dir)
{
    return texcube_depth_nearest(cube, n, v3_(dir[0], dir[1], dir[2]));
}
ORC_API void orc_tex2d_rg_linear_repeat(const uint16_t *img, int w, int h, float s, float t, float *out2)
{
    tex2d_rg_linear_repeat(img, w, h, s, t, &out2[0], &out2[1]);
}
ORC_API void orc_equirect_uv(const float *dir, float *out2)
{
    v3 d = normalize3(v3_(dir[0], dir[1], dir[2]));
    out2[0] = fmaf(atan2f(d.z, d.x), 0.1591f, 0.5f);
    out2[1] = 1.0f - fmaf(asinf(d.y), 0.3183f, 0.5f);
}
ORC_API int orc_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* Geometry pass: gpass.vert / gpass.frag over the frame's draw calls (VulkanRenderer.cs:378-404),
 * BACK faces culled, CCW front, depth test LESS, four RGBA16F targets cleared to
 * (0,0,0,0), (0,0,0,1), (0,0,0,1), (0,0,0,1) (VulkanRenderer.cs:1155-1161; GraphicsPipeline.cs:71-73).
 * Visibility is the rasteriser's and the depth test's: per pixel the covering front-facing
 * triangle with the smallest depth z/w, the earlier draw (then the earlier triangle) winning a
 * tie, as LESS does in submission order.  Coverage as in orc_light_coverage; vertex outputs are
 * interpolated perspective-correctly, sum(Ei*ai)/sum(Ei), and rounded to binary32.
 * Material textures are single-level RGBA8 images read through textureSampler: LINEAR, REPEAT
 * (its anisotropic filtering is implementation-defined and not modelled); albedo and the
 * ao/roughness/metalness image are _SRGB, the normal map _UNORM (Texture.cs:21, Material.cs:20-22). */

typedef struct { const uint8_t *rgba8; int32_t width, height; } orc_texture;

typedef struct {
    const float *positions, *texcoords, *normals, *tangents;  /* 3, 2, 3, 3 floats per vertex */
    int32_t n_verts;
    const uint32_t *indices; int32_t n_tris;
    float model[16];
    orc_texture albedo, normal, arm;
} orc_gpass_draw;

typedef struct {
    const orc_gpass_draw *draws; int32_t n_draws;
    const orc_scene_info *scene;
    int32_t width, rows, frame_rows, row0;
    uint16_t *albedo, *normal, *arm, *position;   /* RGBA16F, rows * width texels each */
} orc_gpass_args;

typedef struct { float pos[4], uv[2], tbn[9], clip[4]; } gp_vertex;

/* gpass.vert:24-35 */
static void gpass_vertex(const orc_gpass_draw *d, const float *vp, uint32_t i, gp_vertex *o)
{
    const float p4[4] = {d->positions[3 * i], d->positions[3 * i + 1], d->positions[3 * i + 2], 1.0f};
    mat4_vec4(d->model, p4, o->pos);
    mat4_vec4(vp, o->pos, o->clip);   /* (proj * view) * outPosition */
    o->uv[0] = d->texcoords[2 * i]; o->uv[1] = d->texcoords[2 * i + 1];
    const float t4[4] = {d->tangents[3 * i], d->tangents[3 * i + 1], d->tangents[3 * i + 2], 0.0f};
    const float n4[4] = {d->normals[3 * i], d->normals[3 * i + 1], d->normals[3 * i + 2], 0.0f};
    float tw[4], nw[4];
    mat4_vec4(d->model, t4, tw);
    mat4_vec4(d->model, n4, nw);
    v3 T = normalize3(v3_(tw[0], tw[1], tw[2]));
    const v3 N = normalize3(v3_(nw[0], nw[1], nw[2]));
    const float tn = dot3(T, N);
    T = normalize3(v3_(T.x - tn * N.x, T.y - tn * N.y, T.z - tn * N.z));
    const v3 B = v3_(fmaf(N.y, T.z, -(N.z * T.y)), fmaf(N.z, T.x, -(N.x * T.z)), fmaf(N.x, T.y, -(N.y * T.x)));
    o->tbn[0] = T.x; o->tbn[1] = T.y; o->tbn[2] = T.z;
    o->tbn[3] = B.x; o->tbn[4] = B.y; o->tbn[5] = B.z;
    o->tbn[6] = N.x; o->tbn[7] = N.y; o->tbn[8] = N.z;
}

/* textureSampler on a single-level RGBA8 image: LINEAR, REPEAT; srgb: decode r,g,b before filtering */
static void tex2d_rgba8_linear_repeat(const orc_texture *t, int srgb, float s, float tt, float *o4)
{
    const int w = t->width, h = t->height;
    float x = s * (float)w - 0.5f, y = tt * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float al = x - fx, be = y - fy;
    int i0 = wrap_i((int)fx, w, 0), i1 = wrap_i((int)fx + 1, w, 0);
    int j0 = wrap_i((int)fy, h, 0), j1 = wrap_i((int)fy + 1, h, 0);
    float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
    const uint8_t *p00 = t->rgba8 + ((size_t)j0 * w + i0) * 4, *p10 = t->rgba8 + ((size_t)j0 * w + i1) * 4;
    const uint8_t *p01 = t->rgba8 + ((size_t)j1 * w + i0) * 4, *p11 = t->rgba8 + ((size_t)j1 * w + i1) * 4;
    for (int c = 0; c < 4; c++) {
        float a, b, cc, d;
        if (srgb && c < 3) { a = SRGB_LUT[p00[c]]; b = SRGB_LUT[p10[c]]; cc = SRGB_LUT[p01[c]]; d = SRGB_LUT[p11[c]]; }
        else { a = (float)p00[c] * (1.0f / 255.0f); b = (float)p10[c] * (1.0f / 255.0f); cc = (float)p01[c] * (1.0f / 255.0f); d = (float)p11[c] * (1.0f / 255.0f); }
        o4[c] = fmaf(w11, d, fmaf(w01, cc, fmaf(w10, b, w00 * a)));
    }
}

/* gpass.frag:18-28 */
static void gpass_fragment(const orc_gpass_draw *d, float u, float vv, const float *tbn, float *alb, float *wn, float *arm)
{
    float nm[4];
    tex2d_rgba8_linear_repeat(&d->albedo, 1, u, vv, alb);
    tex2d_rgba8_linear_repeat(&d->normal, 0, u, vv, nm);
    tex2d_rgba8_linear_repeat(&d->arm, 1, u, vv, arm);
    const v3 n = normalize3(v3_(fmaf(nm[0], 2.0f, -1.0f), fmaf(nm[1], 2.0f, -1.0f), fmaf(nm[2], 2.0f, -1.0f)));
    /* mat3(T, B, N) * n */
    wn[0] = fmaf(tbn[6], n.z, fmaf(tbn[3], n.y, tbn[0] * n.x));
    wn[1] = fmaf(tbn[7], n.z, fmaf(tbn[4], n.y, tbn[1] * n.x));
    wn[2] = fmaf(tbn[8], n.z, fmaf(tbn[5], n.y, tbn[2] * n.x));
}

/* probes for the golden vectors recorded from gpass.vert.spv / gpass.frag.spv */
ORC_API void orc_gpass_vertex(const orc_scene_info *scene, const orc_gpass_draw *d, uint32_t i, float *pos4, float *clip4, float *tbn9)
{
    float vp[16];
    gp_vertex o;
    mat4_mul(scene->camera_view, scene->camera_proj, vp);
    gpass_vertex(d, vp, i, &o);
    memcpy(pos4, o.pos, 16); memcpy(clip4, o.clip, 16); memcpy(tbn9, o.tbn, 36);
}
ORC_API void orc_gpass_fragment(const orc_gpass_draw *d, float u, float v, const float *tbn9, float *alb4, float *wn3, float *arm4)
{
    build_srgb_lut();
    gpass_fragment(d, u, v, tbn9, alb4, wn3, arm4);
}

ORC_API void orc_gbuffer(const orc_gpass_args *a)
{
    build_srgb_lut();
    const size_t npx = (size_t)a->width * a->rows;
    uint64_t *vis = (uint64_t *)malloc(npx * sizeof(uint64_t));   /* depth bits << 32 | triangle number */
    for (size_t i = 0; i < npx; i++) vis[i] = ~0ull;
    float vp[16];
    mat4_mul(a->scene->camera_view, a->scene->camera_proj, vp);
    const double sx = 2.0 / (double)a->width, sy = 2.0 / (double)a->frame_rows;
    /* pass 1: visibility */
    uint32_t base = 0;
    for (int di = 0; di < a->n_draws; di++) {
        const orc_gpass_draw *d = a->draws + di;
        for (int t = 0; t < d->n_tris; t++) {
            gp_vertex v[3];
            float clip[3][4];
            for (int k = 0; k < 3; k++) { gpass_vertex(d, vp, d->indices[3 * t + k], &v[k]); memcpy(clip[k], v[k].clip, sizeof(clip[k])); }
            cov_tri tr;
            cov_setup(clip[0], clip[1], clip[2], &tr);
            if (!(tr.D < 0.0)) continue;            /* BACK faces (D > 0) and degenerate triangles are culled */
            for (int i = 0; i < 3; i++) { tr.a[i] = -tr.a[i]; tr.b[i] = -tr.b[i]; tr.c[i] = -tr.c[i]; }
            tr.D = -tr.D;
            int bx0, by0, bx1, by1;
            /* the bounding box helper works in texels of a square image: use it per axis */
            {
                int qx0, qy0, qx1, qy1, rx0, ry0, rx1, ry1;
                if (!shadow_bbox(clip, a->width, &qx0, &qy0, &qx1, &qy1)) continue;
                if (!shadow_bbox(clip, a->frame_rows, &rx0, &ry0, &rx1, &ry1)) continue;
                bx0 = qx0; bx1 = qx1; by0 = ry0 - a->row0; by1 = ry1 - a->row0;
                if (by0 < 0) by0 = 0;
                if (by1 > a->rows - 1) by1 = a->rows - 1;
                if (by0 > by1) continue;
            }
            const double inv_D = 1.0 / tr.D;  /* one division per triangle: depth = z * (1/w) */
            for (int y = by0; y <= by1; y++) {
                const double yn = fma((double)(a->row0 + y) + 0.5, sy, -1.0);
                for (int x = bx0; x <= bx1; x++) {
                    const double xn = fma((double)x + 0.5, sx, -1.0);
                    const double e0 = cov_edge(&tr, 0, xn, yn), e1 = cov_edge(&tr, 1, xn, yn), e2 = cov_edge(&tr, 2, xn, yn);
                    if (!(e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0 && (e0 + e1) + e2 > 0.0)) continue;
                    const double zs = fma(e2, tr.z[2], fma(e1, tr.z[1], e0 * tr.z[0]));
                    if (!(zs >= 0.0 && zs <= tr.D)) continue;
                    const float depth = (float)(zs * inv_D) + 0.0f;
                    if (!(depth < 1.0f)) continue;   /* CompareOp.Less against the cleared depth 1.0 (GraphicsPipeline.cs:71-73) */
                    uint32_t bits; memcpy(&bits, &depth, 4);
                    const uint64_t key = ((uint64_t)bits << 32) | (uint64_t)(base + (uint32_t)t);
                    uint64_t *o = vis + (size_t)y * a->width + x;
                    if (key < *o) *o = key;
                }
            }
        }
        base += (uint32_t)d->n_tris;
    }
    /* pass 2: shade the visible triangle of every pixel (gpass.frag) */
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < a->rows; y++)
        for (int x = 0; x < a->width; x++) {
            const size_t p = (size_t)y * a->width + x;
            uint16_t *oa = a->albedo + p * 4, *on = a->normal + p * 4, *om = a->arm + p * 4, *op = a->position + p * 4;
            if (vis[p] == ~0ull) { /* clear values */
                oa[0] = oa[1] = oa[2] = oa[3] = 0;
                on[0] = on[1] = on[2] = 0; on[3] = 0x3C00;
                om[0] = om[1] = om[2] = 0; om[3] = 0x3C00;
                op[0] = op[1] = op[2] = 0; op[3] = 0x3C00;
                continue;
            }
            uint32_t id = (uint32_t)(vis[p] & 0xFFFFFFFFu), b0 = 0;
            int di = 0;
            while (id - b0 >= (uint32_t)a->draws[di].n_tris) { b0 += (uint32_t)a->draws[di].n_tris; di++; }
            const orc_gpass_draw *d = a->draws + di;
            const int t = (int)(id - b0);
            gp_vertex v[3];
            for (int k = 0; k < 3; k++) gpass_vertex(d, vp, d->indices[3 * t + k], &v[k]);
            cov_tri tr;
            cov_setup(v[0].clip, v[1].clip, v[2].clip, &tr);
            for (int i = 0; i < 3; i++) { tr.a[i] = -tr.a[i]; tr.b[i] = -tr.b[i]; tr.c[i] = -tr.c[i]; }
            const double xn = fma((double)x + 0.5, sx, -1.0), yn = fma((double)(a->row0 + y) + 0.5, sy, -1.0);
            const double e0 = cov_edge(&tr, 0, xn, yn), e1 = cov_edge(&tr, 1, xn, yn), e2 = cov_edge(&tr, 2, xn, yn);
            /* attributes are interpolated in binary32, as in the shadow pass: the binary64 edge values rounded to binary32
             * are the perspective-correct weights, sum(Ei * Ai) * (1 / sum(Ei)) */
            const float f0 = (float)e0, f1 = (float)e1, f2 = (float)e2;
            const float inv_s = rcpf_((f0 + f1) + f2);  /* one division per pixel */
#define GP_LERP(field, k) (fmaf(f2, v[2].field[k], fmaf(f1, v[1].field[k], f0 * v[0].field[k])) * inv_s)
            const float u = GP_LERP(uv, 0), vv = GP_LERP(uv, 1);
            float pos[4], tbn[9];
            for (int k = 0; k < 4; k++) pos[k] = GP_LERP(pos, k);
            for (int k = 0; k < 9; k++) tbn[k] = GP_LERP(tbn, k);
#undef GP_LERP
            float alb[4], arm[4], wn[3];
            gpass_fragment(d, u, vv, tbn, alb, wn, arm);
            for (int k = 0; k < 4; k++) { oa[k] = f2h(alb[k]); om[k] = f2h(arm[k]); op[k] = f2h(pos[k]); }
            for (int k = 0; k < 3; k++) on[k] = f2h(wn[k]);
            on[3] = 0x3C00;
        }
    free(vis);
}

/* ------------------------------------------------------------------------------------------ */
/* Skybox draw of the composition pass (VulkanRenderer.cs:436-443; cubemap.vert, skybox.frag): a unit
 * cube around the camera, view matrix reduced to its rotation, colour = texture(skybox, position
 * on the cube).  The cube lookup depends only on the direction of that position, which is the
 * pixel's view ray: ray_view = (xn / proj[0][0], yn / proj[1][1], -1) for the reference's
 * symmetric perspective (Camera.cs:30-35), turned into world space by the transposed rotation
 * of the view matrix.  Written where the composition draw later discards (alpha == 0). */
typedef struct {
    const uint16_t *env; int32_t env_n;     /* RGBA16F cube, row-major faces */
    const orc_scene_info *scene;
    int32_t width, rows, frame_rows, row0;
    uint16_t *out;                          /* RGBA16F rows * width */
} orc_skybox_args;

ORC_API void orc_skybox(const orc_skybox_args *a)
{
    const float *V = a->scene->camera_view, *P = a->scene->camera_proj;
    const float sx = 2.0f / (float)a->width, sy = 2.0f / (float)a->frame_rows;
    const float ipx = rcpf_(P[0]), ipy = rcpf_(P[5]);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < a->rows; y++)
        for (int x = 0; x < a->width; x++) {
            const float xn = fmaf((float)x + 0.5f, sx, -1.0f), yn = fmaf((float)(a->row0 + y) + 0.5f, sy, -1.0f);
            const float vx = xn * ipx, vy = yn * ipy, vz = -1.0f;
            const v3 d = v3_(fmaf(vz, V[2], fmaf(vy, V[1], vx * V[0])), fmaf(vz, V[6], fmaf(vy, V[5], vx * V[4])), fmaf(vz, V[10], fmaf(vy, V[9], vx * V[8])));
            const v3 c = texcube_rgb_linear(a->env, a->env_n, d);
            uint16_t *o = a->out + ((size_t)y * a->width + x) * 4;
            o[0] = f2h(c.x); o[1] = f2h(c.y); o[2] = f2h(c.z); o[3] = 0x3C00;
        }
}
