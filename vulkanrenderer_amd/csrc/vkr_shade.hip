// Composition pass on gfx950: the fullscreen composition.frag draw and every additive light.frag
// draw of VulkanRenderer.cs:445-498 in one kernel.  A pixel's G-buffer texels are read once and
// held in registers; the two RGBA16F render targets are accumulated in registers and rounded to
// binary16 after every light, exactly as the per-draw additive blend of
// GraphicsPipelineBuilder.cs:157-187 does, then written once.
//
// Work layout: one wave64 per workgroup, shading an 8x8-pixel block, so the 64 lanes of a wave look
// up neighbouring shadow-map texels (one-wave workgroups measured 3 % faster than 16x16-pixel
// groups of four waves, and 20 % faster than 32x32 groups of sixteen).  Light parameters are
// wave-uniform and come in through scalar loads.  The 20 cube-shadow taps of light.frag:131-137 only ever combine
// three x, three y and three z coordinates (v, v - r, v + r), so their face coordinates are
// computed once per coordinate pair instead of once per tap; when every tap of every lane of the
// wave falls on one cube face (the common case) the face selection disappears altogether.
#include <type_traits>
#include "vkr_device.h"
#include "vkr_fastmath.h"
#include "vkr_host.h"

namespace vkr {

static constexpr float PI_F = 3.14159265359f;
static constexpr float INV_PI = 1.0f / PI_F;
static constexpr float LUM_R = 0.2126f, LUM_G = 0.7152f, LUM_B = 0.0722f;

struct ShadeParams {
    int width, rows, tiles_x, tiles_y;
    const uint2 *albedo, *normal, *arm, *position, *background;
    const float *lights;  // VkrLightInfo array viewed as floats, 23 floats per light
    int n_lights;
    int light_begin, light_end;  // the lights this launch draws; a launch with light_begin > 0 continues from the stored targets
    const float *dir_shadow; int dir_res;
    const float *cube_shadow; int cube_res;
    const void *irradiance; int irr_res;
    const void *prefilter; int pref_res;
    const uint32_t *brdf_lut; int lut_res;
    uint2 *out_color, *out_threshold;
    const unsigned long long *coverage;  // per (light, 8x8 tile) masks, or null
    const unsigned *coverage_flags;      // with masks: word n_lights = number of lights that do not cover every pixel; or null
    const float2 *cube_ranges;           // depth ranges of blocks of the cube maps (see cube_ranges_* below), or null
    unsigned range_cells;                // ceil(cube_res / RANGE_CELL)
    const unsigned long long *light_classes;  // per (tile, 64 lights) two masks written by light_classes_kernel, or null
    int class_words;                     // ceil(n_lights / 64)
    float cam[3];
    float dir_dir[3];
    float dir_color[3];
    float light_space[16];
};

// composition.frag:113-123 with a2 = roughness^4; RCP is the reciprocal to use
template <typename RCP>
VKR_DEV float normal_distribution(float NdotH_raw, float a2, RCP rcp)
{
    float NdotH = max0(NdotH_raw);
    float NdotH2 = NdotH * NdotH;
    float denom = fma_(NdotH2, a2 - 1.0f, 1.0f);
    denom = fma_(PI_F * denom, denom, 0.0001f);
    return a2 * rcp(denom);
}

// composition.frag:125-130 with k = (roughness+1)^2/8 and omk = 1 - k precomputed
template <typename RCP>
VKR_DEV float geometry_schlick(float NdotX, float k, float omk, RCP rcp) { return NdotX * rcp(fma_(NdotX, omk, k)); }

// composition.frag:141-143 with omF0 = 1 - F0 precomputed
VKR_DEV f3 fresnel_schlick(float cosTheta, f3 F0, f3 omF0)
{
    float p = pow5(clamp01(1.0f - cosTheta));
    return mk3(fma_(omF0.x, p, F0.x), fma_(omF0.y, p, F0.y), fma_(omF0.z, p, F0.z));
}

VKR_DEV float luminance(float r, float g, float b) { return fma_(b, LUM_B, fma_(g, LUM_G, r * LUM_R)); }

// shadowSampler on the directional map: NEAREST, CLAMP_TO_BORDER with border depth 1.0
VKR_DEV float dir_depth(const float *img, int res, float s, float t)
{
    float fres = (float)res;
    float fi = __builtin_floorf(s * fres), fj = __builtin_floorf(t * fres);
    bool inside = fi >= 0.0f && fi < fres && fj >= 0.0f && fj < fres;
    int i = inside ? (int)fi : 0, j = inside ? (int)fj : 0;
    float d = img[j * res + i];
    return inside ? d : 1.0f;
}

// composition.frag:145-168
VKR_DEV float dir_shadow(const ShadeParams &p, f3 pos)
{
    float q[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        q[j] = fma_(pos.z, p.light_space[8 + j], fma_(pos.y, p.light_space[4 + j], fma_(pos.x, p.light_space[j], p.light_space[12 + j])));
    float iw = rcp_rn(q[3]);
    float px = q[0] * iw, py = q[1] * iw, pz = q[2] * iw;
    float u = fma_(px, 0.5f, 0.5f), v = fma_(py, 0.5f, 0.5f);
    float texel = rcp_rn((float)p.dir_res);
    float shadow = 0.0f;
#pragma unroll
    for (int x = -1; x <= 1; x++)
#pragma unroll
        for (int y = -1; y <= 1; y++) {
            float d = dir_depth(p.dir_shadow, p.dir_res, fma_((float)x, texel, u), fma_((float)y, texel, v));
            shadow += pz > d ? 1.0f : 0.0f;
        }
    shadow = shadow * (1.0f / 9.0f);
    if (pz > 1.0f) shadow = 0.0f;
    return shadow;
}

// textureSampler on the RG16F BRDF LUT: LINEAR, REPEAT on both axes
VKR_DEV void lut_sample(const uint32_t *lut, int n, float s, float t, float &o0, float &o1)
{
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    float al = x - fx, be = y - fy;
    int i0 = wrap_repeat((int)fx, n), i1 = wrap_repeat((int)fx + 1, n);
    int j0 = wrap_repeat((int)fy, n), j1 = wrap_repeat((int)fy + 1, n);
    float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
    half2_t p00 = __builtin_bit_cast(half2_t, lut[j0 * n + i0]), p10 = __builtin_bit_cast(half2_t, lut[j0 * n + i1]);
    half2_t p01 = __builtin_bit_cast(half2_t, lut[j1 * n + i0]), p11 = __builtin_bit_cast(half2_t, lut[j1 * n + i1]);
    o0 = fma_(w11, (float)p11.x, fma_(w01, (float)p01.x, fma_(w10, (float)p10.x, w00 * (float)p00.x)));
    o1 = fma_(w11, (float)p11.y, fma_(w01, (float)p01.y, fma_(w10, (float)p10.y, w00 * (float)p00.y)));
}

// Byte offset of texel (i, j) inside one face = col_bytes(i) + row_bytes(j).  Row-major:
// 4i + 4*res*j.  Tiled (VKR_CUBE_LAYOUT_TILED_8X4): the face is cut into tiles of 8x4 texels,
// one 128-byte cache line each, stored column by column of tiles; inside a tile rows of 8 texels follow
// each other.  With this order texel (i, j) lies at (i>>3) * 32 res + 32 j + 4 (i&7) bytes: linear in j, so a
// row coordinate needs no address arithmetic at all (sixteen slow-class instructions per light less than with
// tiles stored row by row).  A wave's 8x8-pixel footprint then touches about 40 % fewer lines.
template <bool TILED, bool POW2> VKR_DEV unsigned col_bytes(int i, unsigned res)
{
    if (!TILED) return (unsigned)i * 4u;
    // tiled: (i>>3) * (res/4 tiles * 128 B) + (i&7)*4.  With res = 2^k that is bits 0..2 of i moved to bits 2..4 and
    // its bits 3.. moved to bits k+5..: i * (4 + 4 res) holds a copy of i at bit 2 and one at bit k+2 (they do not
    // overlap: i < res), and a mask keeps the wanted part of each -- two instructions instead of three.
    if (POW2) return __umul24((unsigned)i, 4u + 4u * res) & (0x1Cu | ~(32u * res - 1u));
    return __umul24((unsigned)i >> 3, 32u * res - 32u) + (unsigned)i * 4u;
}
template <bool TILED> VKR_DEV unsigned row_bytes(int j, unsigned res)
{
    // tiled: the tiles of a column follow each other and a tile is four rows of 32 bytes, so rows are 32 bytes apart
    return TILED ? (unsigned)j * 32u : __umul24((unsigned)j, res * 4u);
}

// light.frag:20-27, as indices into {v - r, v, v + r}
#define VKR_GRID20A(T)                                                                             \
    T(2, 2, 2) T(2, 0, 2) T(0, 0, 2) T(0, 2, 2) T(2, 2, 0) T(2, 0, 0) T(0, 0, 0) T(0, 2, 0)        \
    T(2, 2, 1) T(2, 0, 1)
#define VKR_GRID20B(T)                                                                             \
    T(0, 0, 1) T(0, 2, 1) T(2, 1, 2) T(0, 1, 2) T(2, 1, 0) T(0, 1, 0)        \
    T(1, 2, 2) T(1, 0, 2) T(1, 0, 0) T(1, 2, 0)
#define VKR_GRID20(T) VKR_GRID20A(T) VKR_GRID20B(T)

// Texel coordinate (already clamped) of face coordinate c / ma.  POW2: the face size is a power
// of two, so scaling by it is exact and (q * 0.5 + 0.5) * res == q * (res/2) + res/2 bit for bit.
template <bool POW2>
VKR_DEV int face_texel(float c, float ima, float fres, int res)
{
    const float q = c * ima;
    const float scaled = POW2 ? fma_(q, 0.5f * fres, 0.5f * fres) : fma_(q, 0.5f, 0.5f) * fres;
    int i, r;
    // floor + convert in one instruction, clamp to [0, res-1] in one instruction
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(scaled));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(res - 1));
    return r;
}

// face_texel for a coordinate known to satisfy |c| <= |major| * (1 - 2^-21): the scaled value then lies
// in (0, res) after rounding, so the clamp of the general rule never acts and is left out.
template <bool POW2>
VKR_DEV int face_texel_inside(float c, float ima, float fres)
{
    const float q = c * ima;
    const float scaled = POW2 ? fma_(q, 0.5f * fres, 0.5f * fres) : fma_(q, 0.5f, 0.5f) * fres;
    int i;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(scaled));
    return i;
}

// 1.0 where cur > closest, else 0.0, as one FP32 instruction: cur_s = cur * 2^60 (exact).  cur is a
// distance minus 0.15, so a positive cur is at least 2^-27 and a non-zero cur - closest at least 2^-51 in
// magnitude: times 2^60 it saturates the [0, 1] clamp of the instruction; zero, negative and NaN give 0.
VKR_DEV float step_closer(float closest, float cur_s)
{
    float t;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(closest), "s"(-0x1p60f), "v"(cur_s));
    return t;
}

// Number of the 20 taps whose stored depth is closer than `cur` when every tap of every lane of
// the wave lies on one cube face of axis AXIS (`face_base` points at it).  The index triple of a tap is
// permuted so that (iu, iw, im) index the coordinates that become sc and tc and the major coordinate.
template <int AXIS, bool TILED, bool POW2, bool X25>
VKR_DEV float taps_one_face(const float *face_base, int res, const float (&X)[3], const float (&Y)[3], const float (&Z)[3], float cur)
{
    const float fres = (float)res;
    const float (&M)[3] = AXIS == 2 ? Z : (AXIS == 1 ? Y : X);
    float R[3];  // RN(1 / major coordinate), signed: dividing by it applies the sign rule of the face coordinates
#pragma unroll
    for (int k = 0; k < 3; k++) R[k] = fast_rcp_rn(M[k]);
    // z major: sc = x/z, tc = -y/|z|;  y major: sc = x/|y|, tc = z/y;  x major: sc = -z/x, tc = -y/|x|
    // column offset in bytes, row number (rows are 32 bytes apart in the tiled layout, res * 4 in the row-major
    // one); [index into U / W][index into M]
    unsigned I[3][3], J[3][3];
    const unsigned row_unit = TILED ? 32u : (unsigned)res * 4u;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int m = 0; m < 3; m++) {
            if (a == 1 && m == 1) { I[a][m] = 0; J[a][m] = 0; continue; }  // no tap uses (v, v)
            const float rs = R[m], ru = __builtin_fabsf(R[m]);
            float u, ur, w, wr;
            if (AXIS == 2) { u = X[a]; ur = rs; w = -Y[a]; wr = ru; }
            else if (AXIS == 1) { u = X[a]; ur = ru; w = Z[a]; wr = rs; }
            else { u = -Z[a]; ur = rs; w = -Y[a]; wr = ru; }
            I[a][m] = col_bytes<TILED, POW2>(face_texel_inside<POW2>(u, ur, fres), (unsigned)res);
            J[a][m] = (unsigned)face_texel_inside<POW2>(w, wr, fres);  // rows are one row_unit apart in both layouts
        }
    const float cur_s = cur * 0x1p60f;
    float count = 0.0f;
    const char *base = (const char *)face_base;
#define VKR_TAP(gx, gy, gz)                                                              \
    {                                                                                    \
        constexpr int iu = AXIS == 0 ? gz : gx, iw = AXIS == 1 ? gz : gy, im = AXIS == 2 ? gz : (AXIS == 1 ? gy : gx); \
        unsigned off;                                                                    \
        if (TILED) asm("v_lshl_add_u32 %0, %1, 5, %2" : "=v"(off) : "v"(J[iw][im]), "v"(I[iu][im])); \
        else off = __umul24(J[iw][im], row_unit) + I[iu][im];                            \
        const float stored = *(const float *)(base + off);                               \
        const float closest = X25 ? stored : stored * 25.0f;  /* X25: the map holds the product */ \
        count += step_closer(closest, cur_s);                                            \
    }
    VKR_GRID20A(VKR_TAP)
    const float count_a = count;  // two independent chains of ten adds instead of one of twenty
    count = 0.0f;
    VKR_GRID20B(VKR_TAP)
#undef VKR_TAP
    return count + count_a;  // an integer in [0, 20], exact in binary32
}

// Depth ranges of the cube maps: a coarse image beside each face from which light_classes_kernel settles most lights of
// most 8x8-pixel tiles with one fetch, where the shader reads twenty texels per pixel.  Level k (0..RANGE_LEVELS-1) holds,
// for every block of 2S x 2S texels at stride S = RANGE_CELL << k in both directions (block (bx, by) covers columns
// [bx S, bx S + 2S) and rows [by S, by S + 2S), cut at the face's edge), the smallest and the largest of the values the
// taps compare against (25 x depth, rounded as the taps round it).  Blocks overlap by half, so a window of texels
// narrower than S lies inside the block that holds its first column and row: with i0 the first column and bx = i0 / S,
// the last column is below i0 + S < (bx + 2) S.  A block with a NaN in it holds (-inf, +inf) and settles nothing.
// Memory: [light][slab][face][cells][cells] pairs of floats with cells = ceil(res / RANGE_CELL); slab k < RANGE_LEVELS is
// level k (rows and slabs of the coarser levels are padded to the size of level 0, which keeps the address arithmetic
// uniform; the padding is (-inf, +inf) and never read), slab RANGE_LEVELS the cells the levels are built from.
static constexpr int RANGE_CELL = 16, RANGE_LEVELS = 4;

#ifdef VKR_EXPERIMENT_COUNT  // kernel experiments only: what happens to each (wave, light) pair
__device__ unsigned long long shade_counts[8];
#define VKR_COUNT(i) do { if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) atomicAdd(&shade_counts[i], 1ull); } while (0)
#else
#define VKR_COUNT(i) do { } while (0)
#endif

// The count for a wave that straddles a cube edge or corner: the face is chosen per tap and per lane
// (light.frag:133 through shadowSampler: NEAREST on the face the direction's major axis selects; ties go
// to z over y over x).  One axis after the other gets the column / row tables of the one-face path (with
// the clamp of the general rule and the face's base offset folded into the columns), so only one set of
// tables is live; the twenty byte offsets start as those of the x faces, are replaced per lane and tap
// by those of the y faces where |y| >= |x|, then by those of the z faces where |z| >= |x| and |z| >= |y|,
// which is the selection rule.  All twenty loads are then in flight together as in the one-face path: the
// memory latency, not the instruction count, is what a tap-by-tap formulation waits for.  An axis that is
// the major one for no tap of the wave (two of three in a wave on one edge) is skipped: its offsets would
// be replaced or never selected.  Dividing by the signed major coordinate applies the sign rule of the
// face coordinates: sc = x/z, -z/x or x/|y|, tc = -y/|z|, -y/|x| or z/y ("c >= 0 ? v : -v" and v * sign(c)
// differ only for c = -0, which can be the major coordinate only when the whole direction is zero).
template <int AXIS, bool TILED, bool POW2, typename EMIT>
VKR_DEV void tap_offsets_of_axis(int res, unsigned face_shift, const float (&X)[3], const float (&Y)[3], const float (&Z)[3], EMIT emit)
{
    const float fres = (float)res;
    const float (&M)[3] = AXIS == 2 ? Z : (AXIS == 1 ? Y : X);
    float R[3];        // RN(1 / major coordinate), signed
    unsigned B[3];     // byte offset of the face each major candidate selects
#pragma unroll
    for (int k = 0; k < 3; k++) {
        R[k] = fast_rcp_rn(M[k]);
        const unsigned face = (unsigned)(AXIS * 2) | (__builtin_bit_cast(unsigned, M[k]) >> 31);
        B[k] = POW2 ? face << face_shift : face * ((unsigned)res * (unsigned)res * 4u);
    }
    unsigned I[3][3], J[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int m = 0; m < 3; m++) {
            if (a == 1 && m == 1) { I[a][m] = 0; J[a][m] = 0; continue; }  // no tap uses (v, v)
            const float rs = R[m], ru = __builtin_fabsf(R[m]);
            float u, ur, w, wr;
            if (AXIS == 2) { u = X[a]; ur = rs; w = -Y[a]; wr = ru; }
            else if (AXIS == 1) { u = X[a]; ur = ru; w = Z[a]; wr = rs; }
            else { u = -Z[a]; ur = rs; w = -Y[a]; wr = ru; }
            I[a][m] = col_bytes<TILED, POW2>(face_texel<POW2>(u, ur, fres, res), (unsigned)res) + B[m];
            J[a][m] = row_bytes<TILED>(face_texel<POW2>(w, wr, fres, res), (unsigned)res);
        }
    int k = 0;
#define VKR_TAP(gx, gy, gz)                                                              \
    {                                                                                    \
        constexpr int iu = AXIS == 0 ? gz : gx, iw = AXIS == 1 ? gz : gy, im = AXIS == 2 ? gz : (AXIS == 1 ? gy : gx); \
        emit(k++, gx, gy, gz, J[iw][im] + I[iu][im]);                                    \
    }
    VKR_GRID20(VKR_TAP)
#undef VKR_TAP
}

template <bool TILED, bool POW2, bool X25>
VKR_DEV float taps_any_face(const float *cube, int res, const float (&X)[3], const float (&Y)[3], const float (&Z)[3], float cur,
                            bool has_x, bool has_y, bool has_z)
{
    const unsigned face_shift = 2u * (unsigned)__builtin_ctz((unsigned)res) + 2u;  // POW2: bytes per face = 1 << face_shift
    const char *base = (const char *)cube;
    unsigned off[20];
#pragma unroll
    for (int k = 0; k < 20; k++) off[k] = 0u;
    if (has_x)
        tap_offsets_of_axis<0, TILED, POW2>(res, face_shift, X, Y, Z, [&](int k, int, int, int, unsigned o) { off[k] = o; });
    if (has_y)
        tap_offsets_of_axis<1, TILED, POW2>(res, face_shift, X, Y, Z, [&](int k, int gx, int gy, int, unsigned o) {
            off[k] = __builtin_fabsf(Y[gy]) >= __builtin_fabsf(X[gx]) ? o : off[k];
        });
    if (has_z)
        tap_offsets_of_axis<2, TILED, POW2>(res, face_shift, X, Y, Z, [&](int k, int gx, int gy, int gz, unsigned o) {
            const float az = __builtin_fabsf(Z[gz]);
            off[k] = ((az >= __builtin_fabsf(X[gx])) & (az >= __builtin_fabsf(Y[gy]))) ? o : off[k];
        });
    const float cur_s = cur * 0x1p60f;
    float count_a = 0.0f, count_b = 0.0f;
#pragma unroll
    for (int k = 0; k < 10; k++) count_a += step_closer(X25 ? *(const float *)(base + off[k]) : *(const float *)(base + off[k]) * 25.0f, cur_s);
#pragma unroll
    for (int k = 10; k < 20; k++) count_b += step_closer(X25 ? *(const float *)(base + off[k]) : *(const float *)(base + off[k]) * 25.0f, cur_s);
    return count_a + count_b;
}

// Which axes can be the major one for some tap of some active lane, as bits 0 (x), 1 (y), 2 (z).  An axis
// never is where its largest candidate magnitude |v| + r stays below the smallest |v| - r of another
// axis (which of the two others may differ from lane to lane).
VKR_DEV unsigned wave_major_axes(float fx, float fy, float fz, float r)
{
    const float ax = __builtin_fabsf(fx), ay = __builtin_fabsf(fy), az = __builtin_fabsf(fz);
    const float xhi = ax + r, yhi = ay + r, zhi = az + r, xlo = ax - r, ylo = ay - r, zlo = az - r;
    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
    auto either = [active](bool a, bool b) { return (__builtin_amdgcn_ballot_w64(a) | __builtin_amdgcn_ballot_w64(b)) == active; };
    return (either(xhi < zlo, xhi < ylo) ? 0u : 1u) | (either(yhi < zlo, yhi < xlo) ? 0u : 2u) | (either(zhi < ylo, zhi < xlo) ? 0u : 4u);
}

// Wave votes on a condition that is already a lane mask (HIP's __all / __any go through an integer select
// and a second compare).
VKR_DEV bool wave_all(bool c) { return __builtin_amdgcn_ballot_w64(c) == __builtin_amdgcn_ballot_w64(true); }
VKR_DEV bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// The single cube face that all 27 combinations of {v-r, v, v+r}^3 select in every active lane of the
// wave, or -1 if there is none.  (fx, fy, fz) is the centre direction v, r the disk radius.  The largest
// magnitude of a coordinate's three candidates is |v| + r and, once |v| > r, the smallest is |v| - r, both
// with the very rounding of the candidates themselves; the comparisons are those of the face-selection
// rule (ties: z over y over x) on these extremes, so agreement here implies the per-tap rule picks the
// face for each tap.  The major coordinate must also win by a margin (factor 1 - 2^-20): the one-face path
// then needs no clamp of its texel coordinates (face_texel_inside), and since the other coordinates' maxima
// are at least r > 0 the major one keeps its sign over its candidates and stays far inside fast_rcp_rn's
// domain.  Decided with wave-wide votes: no per-lane selects.
VKR_DEV int wave_common_face(float fx, float fy, float fz, float r)
{
    const float ax = __builtin_fabsf(fx), ay = __builtin_fabsf(fy), az = __builtin_fabsf(fz);
    const float xhi = ax + r, yhi = ay + r, zhi = az + r;
    constexpr float K = 1.0f - 0x1p-20f;
    const float xs = (ax - r) * K, ys = (ay - r) * K, zs = (az - r) * K;
    // each comparison goes into its own ballot and the masks are combined as scalars: a ballot of a combined
    // condition is compiled as a select and a second compare
    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
    auto all2 = [active](bool a, bool b) { return (__builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b)) == active; };
    int axis;
    float major;
    if (all2(zs >= xhi, zs >= yhi)) { axis = 4; major = fz; }        // z major for every combination
    else if (all2(zhi < ys, ys >= xhi)) { axis = 2; major = fy; }    // never z major, always y over x
    else if (all2(zhi < xs, yhi < xs)) { axis = 0; major = fx; }     // never z major, never y over x
    else return -1;
    // beyond 2^100 the fast reciprocal leaves its exact domain (and infinity passes the comparisons above):
    // such waves take the clamped path, which stays inside the cube whatever the coordinates are
    if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(major) < 0x1p100f) != active) return -1;
    const unsigned long long neg = __builtin_amdgcn_ballot_w64(major < 0.0f);
    return neg == 0ull ? axis : (neg == active ? axis + 1 : -1);
}

// Workgroups are dealt to the 8 XCDs round-robin by id, and each XCD has its own L2.  This maps block id -> tile so
// that the workgroups resident on an XCD at one time cover a compact screen region, and therefore a compact region
// of every shadow map, which keeps their shared texels in that L2: tiles are ordered in 64-tile-wide (512-pixel)
// column blocks, row-major inside a block, the order is cut into runs of SHADE_RUN tiles, and the runs are dealt to
// the XCDs in turn.  An XCD's share is thus spread over the whole frame in small pieces: every XCD gets the same mix
// of cheap (background, back-facing) and expensive tiles and they finish together -- with one contiguous eighth of the
// frame per XCD the slowest eighth set the time of a 4K launch, 10 % above the time per pixel of larger frames.
// Returns false for the ids past the last tile (the grid is rounded up to whole rounds of 8 runs).
#ifndef VKR_SHADE_RUN
#define VKR_SHADE_RUN 256
#endif
static constexpr unsigned SHADE_RUN = VKR_SHADE_RUN;
static constexpr int SHADE_LIGHTS_PER_LAUNCH = 64;  // see vkr_shade

VKR_DEV bool tile_of_block(unsigned b, int tiles_x, int tiles_y, int &tx, int &ty)
{
    constexpr unsigned CB = 64u;
    const unsigned total = (unsigned)tiles_x * tiles_y;
    const unsigned xcd = b & 7u, q = b >> 3;                          // the q-th workgroup of this XCD
    const unsigned v = ((q / SHADE_RUN) * 8u + xcd) * SHADE_RUN + q % SHADE_RUN;  // position in the column-block order
    if (v >= total) return false;
    const unsigned block_tiles = CB * (unsigned)tiles_y;
    const unsigned cb = min(v / block_tiles, (unsigned)(tiles_x - 1) / CB);
    const unsigned r = v - cb * block_tiles;
    const unsigned cbw = min(CB, (unsigned)tiles_x - cb * CB);
    ty = (int)(r / cbw);
    tx = (int)(cb * CB + r % cbw);
    return true;
}

VKR_DEV unsigned f16_low_of(float x) { return (unsigned)__builtin_bit_cast(unsigned short, to_f16(x)); }
// The additive blend of one light draw into an RGBA16F target: target + x formed in binary32 (rounded), then rounded to
// binary16.  The target is kept as a binary16 value in the low half of a register: a mixed-precision fma reads it as it
// is (x * 1.0 + target, result in binary32), one conversion follows -- an add and two conversions before.
// (v_fma_mixlo_f16 would do it in one instruction, but it rounds the exact sum once, which differs from the two roundings
// in one result out of 10^4.)
VKR_DEV unsigned blend_f16(unsigned target, float x)
{
    float sum;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[0,0,1]" : "=v"(sum) : "v"(x), "v"(target));
    return f16_low_of(sum);
}

VKR_DEV uint2 pack4_f16(unsigned r, unsigned g, unsigned b)  // alpha 1.0
{
    return make_uint2(__builtin_amdgcn_perm(g, r, 0x05040100u), __builtin_amdgcn_perm(0x3C00u, b, 0x05040100u));
}

// Five waves per SIMD (96 VGPRs, none spilled).  Measured at 4K / 64 lights with this loop: 5 waves 3.58 ms,
// 6 waves (80 VGPRs, values reloaded from scratch memory inside the loop) 3.72 ms; an earlier, longer loop
// measured 4 waves 5.5 ms, 5 waves 4.6, 6 waves 4.5, 7 waves (spills inside the loop) 4.6.
// The loop is bound by VALU issue, and on this chip v_add/mul/fma_f32 issue in about 3 clocks per wave while
// every integer, compare, select, convert and min/max instruction takes a little over 4 and a reciprocal or
// root 8: the code below prefers the former (clamped fmas for comparisons, signed reciprocals instead of
// selects, votes on scalar masks).  The whole kernel is kept under about 30 KB of code: a 36 KB variant
// ran 25 % slower for instruction-cache misses alone.
// MASKED: each light's draw touches only the pixels its light volume covers (one 64-bit mask per
// light and tile, bit = lane); a light that covers none of the wave's pixels is skipped whole.
// LAYOUT: VKR_CUBE_LAYOUT_* of the point-shadow cubes.
// CLASSES: the per-tile light classes of light_classes_kernel are read (p.light_classes).
template <int LAYOUT, bool POW2, bool MASKED, bool CLASSES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void shade_kernel(const ShadeParams p)
{
    constexpr bool TILED = LAYOUT != VKR_CUBE_LAYOUT_LINEAR, X25 = LAYOUT == VKR_CUBE_LAYOUT_TILED_8X4_X25;
    // one wave per workgroup, one 8x8-pixel tile per wave
    const int lane = threadIdx.x;
    int tile_x, tile_y;
    if (!tile_of_block(blockIdx.x, p.tiles_x, p.tiles_y, tile_x, tile_y)) return;
    const int x = tile_x * 8 + (lane & 7);
    const int y = tile_y * 8 + (lane >> 3);
    if (x >= p.width || y >= p.rows) return;
    const size_t px = (size_t)y * p.width + x;

    const rgba al = unpack4(p.albedo[px]);
    const rgba no = unpack4(p.normal[px]);
    const rgba ar = unpack4(p.arm[px]);
    const rgba po = unpack4(p.position[px]);
    const f3 albedo = mk3(al.r, al.g, al.b), nraw = mk3(no.r, no.g, no.b), pos = mk3(po.r, po.g, po.b);
    const float ao = ar.r, rough = ar.g, metal = ar.b;
    const f3 cam = mk3(p.cam[0], p.cam[1], p.cam[2]);
    auto rcp_ieee = [](float v) { return rcp_rn(v); };

    // shared by both shaders: composition.frag:45-54, light.frag:43-51
    const float om = 1.0f - metal;
    const f3 F0 = mk3(fma_(albedo.x, metal, 0.04f * om), fma_(albedo.y, metal, 0.04f * om), fma_(albedo.z, metal, 0.04f * om));
    const f3 omF0 = mk3(1.0f - F0.x, 1.0f - F0.y, 1.0f - F0.z);
    const f3 N = normalize3(nraw);
    const f3 Vraw = sub3(cam, pos);
    const float view_distance = sqrt_rn(dot3(Vraw, Vraw));
    const f3 V = scale3(Vraw, rcp_rn(view_distance));
    const float NdotV = max0(dot3(N, V));
    const float a = rough * rough, a2 = a * a;
    const float r1 = rough + 1.0f, k = (r1 * r1) * 0.125f, omk = 1.0f - k;
    const float g1V = geometry_schlick(NdotV, k, omk, rcp_ieee);
    const bool degenerate_normal = !(dot3(nraw, nraw) > 0.0f);

    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    const bool resume = p.light_begin > 0;  // launch-uniform
    if (resume) {
        // The targets as the launch for the lights before light_begin left them: binary16 values, so reading them
        // back is exact and the chain of per-draw roundings goes on as if it had never been cut.
        const rgba pc = unpack4(p.out_color[px]), pt = unpack4(p.out_threshold[px]);
        c0 = pc.r; c1 = pc.g; c2 = pc.b; t0 = pt.r; t1 = pt.g; t2 = pt.b;
    } else if (p.background) { rgba bg = unpack4(p.background[px]); c0 = bg.r; c1 = bg.g; c2 = bg.b; }

    if (!resume && al.a != 0.0f) {  // composition.frag:39-43
        f3 Lo;
        {
            f3 L = normalize3(neg3(mk3(p.dir_dir[0], p.dir_dir[1], p.dir_dir[2])));
            f3 H = normalize3(add3(V, L));
            float D = normal_distribution(dot3(N, H), a2, rcp_ieee);
            float NdotL = max0(dot3(N, L));
            float G = g1V * geometry_schlick(NdotL, k, omk, rcp_ieee);
            f3 F = fresnel_schlick(max0(dot3(H, V)), F0, omF0);
            f3 kD = mk3((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
            float isd = rcp_rn(fma_(4.0f * NdotV, NdotL, 0.0001f));
            float DG = D * G;
            f3 spec = mk3((DG * F.x) * isd, (DG * F.y) * isd, (DG * F.z) * isd);
            float lit = 1.0f - dir_shadow(p, pos);
            Lo.x = ((fma_(kD.x * albedo.x, INV_PI, spec.x) * p.dir_color[0]) * lit) * NdotL;
            Lo.y = ((fma_(kD.y * albedo.y, INV_PI, spec.y) * p.dir_color[1]) * lit) * NdotL;
            Lo.z = ((fma_(kD.z * albedo.z, INV_PI, spec.z) * p.dir_color[2]) * lit) * NdotL;
        }
        {   // ambient, composition.frag:84-101
            f3 F = fresnel_schlick(NdotV, F0, omF0);
            f3 kD = mk3((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
            f3 irr = texcube_rgb_linear(p.irradiance, p.irr_res, N);
            f3 I = neg3(V);
            float d2 = 2.0f * dot3(N, I);
            f3 R = mk3(fma_(-d2, N.x, I.x), fma_(-d2, N.y, I.y), fma_(-d2, N.z, I.z));
            // every textureLod resolves to mip 0 (sampler MaxLod = 0, one-level cube view)
            f3 pre = texcube_rgb_linear(p.prefilter, p.pref_res, R);
            float bx, by;
            lut_sample(p.brdf_lut, p.lut_res, NdotV, rough, bx, by);
            f3 spec = mk3(pre.x * ((F.x * bx) * by), pre.y * ((F.y * bx) * by), pre.z * ((F.z * bx) * by));
            Lo.x += fma_(kD.x, irr.x * albedo.x, spec.x) * ao;
            Lo.y += fma_(kD.y, irr.y * albedo.y, spec.y) * ao;
            Lo.z += fma_(kD.z, irr.z * albedo.z, spec.z) * ao;
        }
        c0 = Lo.x; c1 = Lo.y; c2 = Lo.z;
        bool bright = luminance(Lo.x, Lo.y, Lo.z) > 1.0f;
        t0 = bright ? Lo.x : 0.0f; t1 = bright ? Lo.y : 0.0f; t2 = bright ? Lo.z : 0.0f;
    }
    // from here on the targets are binary16 values in the low halves of these registers (see blend_f16)
    unsigned h0 = f16_low_of(c0), h1 = f16_low_of(c1), h2 = f16_low_of(c2), g0 = f16_low_of(t0), g1 = f16_low_of(t1), g2 = f16_low_of(t2);

    // A cleared G-buffer texel has normal (0,0,0): the light shader's result for it is 0 under
    // IEEE maxNum semantics, so such pixels skip the light loop.
    if (!degenerate_normal) {
        const float disk = fma_(view_distance, 1.0f / 25.0f, 1.0f) * (1.0f / 25.0f);
        const float fourNdotV = 4.0f * NdotV;
        const size_t face_stride = (size_t)p.cube_res * p.cube_res;
        auto rcp_fast = [](float v) { return fast_rcp_rn(v); };
        // Light records are read-only for the whole launch and their address is wave-uniform: read
        // them through the constant address space so they come in over the scalar cache, and fetch
        // light li + 1 while light li is being shaded.
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat *lights_c = (cfloat *)p.lights;
        f3 lp_next = mk3(0.0f, 0.0f, 0.0f), lc_next = lp_next;
        if (p.light_end > p.light_begin) {
            cfloat *lf = lights_c + p.light_begin * 23;
            lp_next = mk3(lf[16], lf[17], lf[18]);
            lc_next = mk3(lf[20], lf[21], lf[22]);
        }
        typedef const __attribute__((address_space(4))) unsigned long long cmask;
        cmask *cover_c = (cmask *)p.coverage;
        const size_t tiles = (size_t)p.tiles_x * p.tiles_y, tile = (size_t)tile_y * p.tiles_x + tile_x;
        cmask *classes_c = CLASSES ? (cmask *)p.light_classes + tile * (size_t)(p.class_words * 2) : nullptr;
        // The light loop, with or without the per-light masks.  A kernel built for masks holds both forms: when the
        // coverage pass found that every light covers every pixel (the reference's own light radii, with the camera
        // inside every volume) its waves take the form without masks, which costs neither the mask loads nor the
        // selects.  The branch is wave-uniform and the same for every wave of a launch, so only one form is ever
        // in the instruction cache.
        auto light_loop = [&](auto use_masks) {
        constexpr bool MASKS = decltype(use_masks)::value;
        unsigned long long skip_word = 0ull, lit_word = 0ull;  // classes of the 64 lights li belongs to, bit li & 63
        for (int li = p.light_begin; li < p.light_end; li++) {
            const f3 lp = lp_next, lc = lc_next;
            {
                cfloat *lf = lights_c + min(li + 1, p.light_end - 1) * 23;
                lp_next = mk3(lf[16], lf[17], lf[18]);
                lc_next = mk3(lf[20], lf[21], lf[22]);
            }
            if (CLASSES) {
                if (li == p.light_begin || (li & 63) == 0) {
                    skip_word = classes_c[(li >> 6) * 2];
                    lit_word = classes_c[(li >> 6) * 2 + 1];
                }
                if ((skip_word >> (li & 63)) & 1ull) { VKR_COUNT(6); continue; }  // every pixel of the tile has all 20 taps in shadow: adds exactly zero
            }
            const bool tile_lit = (lit_word >> (li & 63)) & 1ull;  // no tap of any pixel of the tile is in shadow
            bool covered = true;
            unsigned long long cover_mask = ~0ull;
            if (MASKS) {
                cover_mask = cover_c[(size_t)li * tiles + tile];  // wave-uniform: a scalar load
                if (cover_mask == 0) continue;
                covered = (cover_mask >> lane) & 1ull;
            }
            // fragToLight = pos - lp; L is its negation normalised (a - b = -(b - a) bit for bit, and the
            // squares in the length are the same)
            const f3 Fv = sub3(pos, lp);
            const float dist2 = dot3(Fv, Fv);
            float dist, inv_dist, att;
            // Lanes outside the domain of the short sequences.  The long and the short forms give the same bits inside
            // it (both round correctly), so the whole wave takes the long one when any lane needs it: a scalar branch,
            // no per-lane one.
            const unsigned long long slow_lanes = __builtin_amdgcn_ballot_w64(!(dist2 >= 0x1p-96f)) | __builtin_amdgcn_ballot_w64(!(dist2 <= 0x1p96f));
            if (__builtin_expect(slow_lanes != 0ull, 0)) {
                dist = __builtin_sqrtf(dist2);
                inv_dist = 1.0f / dist;
                att = 1.0f / (dist * dist);
            } else {  // dist is within [2^-48, 2^48], its square well inside fast_rcp_rn's domain
                dist = fast_sqrt_rn(dist2);
                inv_dist = fast_rcp_rn(dist);
                att = fast_rcp_rn(dist * dist);
            }
            const f3 L = mk3(-Fv.x * inv_dist, -Fv.y * inv_dist, -Fv.z * inv_dist);
            const float NdotL = max0(dot3(N, L));
            // light.frag:121-141
            const float cur = dist - 0.15f;
            const float X[3] = {Fv.x - disk, Fv.x, Fv.x + disk};
            const float Y[3] = {Fv.y - disk, Fv.y, Fv.y + disk};
            const float Z[3] = {Fv.z - disk, Fv.z, Fv.z + disk};
            const float *cube = p.cube_shadow + (size_t)li * 6 * face_stride;
            // The shadow factor multiplies a product whose last factor is N.L; where N.L is zero the result is the
            // same zero whatever the taps say (the other factors are finite inside the fast domain, given finite
            // G-buffer values and light colours), and adding it leaves both targets as they are: a wave that
            // faces away from the light is done.  Masks of single comparisons are combined as scalars.
            unsigned long long live = slow_lanes | __builtin_amdgcn_ballot_w64(NdotL > 0.0f);
            if (MASKS) live &= cover_mask;
            if (live == 0ull) { VKR_COUNT(0); continue; }
            // The rest of the draw once the number of taps in shadow is known; LIT: known to be zero for the whole tile.
            auto finish = [&](auto lit_known, float count) {
            constexpr bool LIT = decltype(lit_known)::value;
            // all 20 taps in shadow: the factor 1 - 20/20 is an exact zero, same argument
            if (!LIT && (live & (slow_lanes | __builtin_amdgcn_ballot_w64(count < 20.0f))) == 0ull) { VKR_COUNT(4); return; }
            VKR_COUNT(5);
            const f3 Hv = add3(V, L);
            const float hlen2 = dot3(Hv, Hv);  // at most about 4: only the lower end of the fast domain needs a test
            float inv_hlen;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(hlen2 >= 0x1p-96f)) != 0ull, 0)) inv_hlen = 1.0f / __builtin_sqrtf(hlen2);  // same bits inside the domain
            else inv_hlen = fast_rcp_rn(fast_sqrt_rn(hlen2));
            const f3 H = scale3(Hv, inv_hlen);
            // the denominators below lie in [1e-4, 8]: well inside fast_rcp_rn's exact domain
            const float D = normal_distribution(dot3(N, H), a2, rcp_fast);
            const float G = g1V * geometry_schlick(NdotL, k, omk, rcp_fast);
            const f3 F = fresnel_schlick(max0(dot3(H, V)), F0, omF0);
            const float isd = fast_rcp_rn(fma_(fourNdotV, NdotL, 0.0001f));
            const float DG = D * G;

            // no tap in shadow: the factor is 1 - 0 * (1/20) = 1 and x * 1 = x bit for bit, so the three products are left out
            const float lit = LIT ? 1.0f : 1.0f - count * (1.0f / 20.0f);

            const float o0 = ((fma_((1.0f - F.x) * albedo.x, INV_PI, (DG * F.x) * isd) * (lc.x * att)) * lit) * NdotL;
            const float o1 = ((fma_((1.0f - F.y) * albedo.y, INV_PI, (DG * F.y) * isd) * (lc.y * att)) * lit) * NdotL;
            const float o2 = ((fma_((1.0f - F.z) * albedo.z, INV_PI, (DG * F.z) * isd) * (lc.z * att)) * lit) * NdotL;
            const bool over = luminance(o0, o1, o2) > 1.0f;  // light.frag:78-82
            const bool bright = covered && over;
            if (MASKS) {  // an uncovered pixel is not part of this draw: its targets keep their bits
                h0 = covered ? blend_f16(h0, o0) : h0; h1 = covered ? blend_f16(h1, o1) : h1; h2 = covered ? blend_f16(h2, o2) : h2;
            } else {
                h0 = blend_f16(h0, o0); h1 = blend_f16(h1, o1); h2 = blend_f16(h2, o2);
            }
            if ((__builtin_amdgcn_ballot_w64(over) & cover_mask) != 0ull) {  // adding 0 to the thresholded target changes nothing
                g0 = blend_f16(g0, bright ? o0 : 0.0f);
                g1 = blend_f16(g1, bright ? o1 : 0.0f);
                g2 = blend_f16(g2, bright ? o2 : 0.0f);
            }
            };
            if (tile_lit) { VKR_COUNT(7); finish(std::true_type{}, 0.0f); continue; }
            const int face0 = wave_common_face(Fv.x, Fv.y, Fv.z, disk);
            float count;
            if (face0 >= 0) {
                VKR_COUNT(2);
                const float *fb = cube + (size_t)face0 * face_stride;
                if (face0 >= 4) count = taps_one_face<2, TILED, POW2, X25>(fb, p.cube_res, X, Y, Z, cur);
                else if (face0 >= 2) count = taps_one_face<1, TILED, POW2, X25>(fb, p.cube_res, X, Y, Z, cur);
                else count = taps_one_face<0, TILED, POW2, X25>(fb, p.cube_res, X, Y, Z, cur);
            } else {  // the wave straddles a cube edge or corner
                VKR_COUNT(3);
                const unsigned axes = wave_major_axes(Fv.x, Fv.y, Fv.z, disk);
                count = taps_any_face<TILED, POW2, X25>(cube, p.cube_res, X, Y, Z, cur, axes & 1u, axes & 2u, axes & 4u);
            }
            finish(std::false_type{}, count);
        }
        };
        if (MASKED) {
            typedef const __attribute__((address_space(4))) unsigned cflag;
            const bool every_light_covers_all = p.coverage_flags != nullptr && ((cflag *)p.coverage_flags)[p.n_lights] == 0u;
            if (every_light_covers_all) light_loop(std::false_type{});
            else light_loop(std::true_type{});
        } else {
            light_loop(std::false_type{});
        }
    }
    p.out_color[px] = pack4_f16(h0, h1, h2);
    p.out_threshold[px] = pack4_f16(g0, g1, g2);
}

// Row-major face -> 8x4-texel tiles.  Eight lanes move one tile (16 bytes each); a wave takes eight tiles side by
// side: it reads four 256-byte row segments and writes eight whole 128-byte lines.
// X25: the tiles hold RN(25 * depth).
template <bool X25>
__global__ __launch_bounds__(256) void tile_depth_kernel(const float4 *src, float4 *dst, int res, unsigned tiles_per_face, unsigned total_tiles)
{
    const unsigned t = (blockIdx.x * 256u + threadIdx.x) >> 3;  // tile index over all faces
    if (t >= total_tiles) return;
    const unsigned q = threadIdx.x & 7u;                        // 16-byte piece of the tile: row q >> 1, half q & 1
    const unsigned face = t / tiles_per_face, tf = t - face * tiles_per_face;
    const unsigned tiles_x = (unsigned)res >> 3;
    const unsigned ty = tf / tiles_x, tx = tf - ty * tiles_x;
    const size_t texel = (size_t)face * res * res + (size_t)(ty * 4u + (q >> 1)) * res + tx * 8u + (q & 1u) * 4u;
    float4 v = src[texel >> 2];
    if (X25) { v.x *= 25.0f; v.y *= 25.0f; v.z *= 25.0f; v.w *= 25.0f; }
    const size_t tile = (size_t)face * tiles_per_face + (size_t)tx * ((unsigned)res >> 2) + ty;  // tiles are stored column by column
    dst[tile * 8u + q] = v;
}

}  // namespace vkr

extern "C" int32_t vkr_relayout_depth_cubes(const float *linear, float *tiled, int32_t n_faces, int32_t res, int32_t layout, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(linear && tiled, "vkr_relayout_depth_cubes: an image is null");
    VKR_REQUIRE(layout == VKR_CUBE_LAYOUT_TILED_8X4 || layout == VKR_CUBE_LAYOUT_TILED_8X4_X25, "vkr_relayout_depth_cubes: layout %d is not a tiled one", layout);
    VKR_REQUIRE(n_faces >= 0 && res > 0 && res % 8 == 0 && res <= 8192, "vkr_relayout_depth_cubes: %d faces of size %d (size must be a multiple of 8, at most 8192)", n_faces, res);
    const size_t bytes = (size_t)n_faces * res * res * 4;
    const char *a = (const char *)linear, *b = (const char *)tiled;
    VKR_REQUIRE(a + bytes <= b || b + bytes <= a, "vkr_relayout_depth_cubes: source and destination overlap");
    const uint64_t tiles_per_face = (uint64_t)(res / 8) * (res / 4), total = tiles_per_face * (uint64_t)n_faces;
    VKR_REQUIRE(total < (1ull << 28), "vkr_relayout_depth_cubes: too many texels for one call");
    if (total == 0) return VKR_OK;
    const dim3 grid((unsigned)((total * 8 + 255) / 256));
    if (layout == VKR_CUBE_LAYOUT_TILED_8X4_X25)
        hipLaunchKernelGGL(tile_depth_kernel<true>, grid, dim3(256), 0, as_stream(stream), (const float4 *)linear, (float4 *)tiled, res, (unsigned)tiles_per_face, (unsigned)total);
    else
        hipLaunchKernelGGL(tile_depth_kernel<false>, grid, dim3(256), 0, as_stream(stream), (const float4 *)linear, (float4 *)tiled, res, (unsigned)tiles_per_face, (unsigned)total);
    return launch_status("vkr_relayout_depth_cubes");
}

extern "C" int32_t vkr_tile_depth_cubes(const float *linear, float *tiled, int32_t n_faces, int32_t res, void *stream)
{
    return vkr_relayout_depth_cubes(linear, tiled, n_faces, res, VKR_CUBE_LAYOUT_TILED_8X4, stream);
}

namespace vkr {

// Depth ranges, step 1: the smallest and largest comparison value of every RANGE_CELL x RANGE_CELL cell of every face.
// Sixteen lanes take a cell (a wave four cells side by side, four times), each lane sixteen texels of a cell in four
// 16-byte loads: in the tiled layouts a cell is two runs of 512 contiguous bytes (four 8x4 tiles of one tile column
// each), in the row-major one sixteen 64-byte row segments.  FAST needs a face size that is a multiple of RANGE_CELL;
// the other form reads texel by texel.  Cells are written to the slab behind the last level.
static constexpr unsigned RANGE_CELLS_PER_WAVE = 16;
template <int LAYOUT, bool FAST>
__global__ __launch_bounds__(256) void cube_range_cells_kernel(const float *cubes, float2 *ranges, int res, unsigned cells, unsigned total_cells)
{
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    float lo[4], hi[4];
    bool nan[4] = {false, false, false, false};
    unsigned cell[4];
#pragma unroll
    for (unsigned it = 0; it < 4; it++) {
        cell[it] = wave * RANGE_CELLS_PER_WAVE + it * 4u + (lane >> 4);  // over all faces: (face_all * cells + cy) * cells + cx
        lo[it] = __builtin_inff(); hi[it] = -__builtin_inff();
        if (cell[it] >= total_cells) continue;
        const unsigned cx = cell[it] % cells, cy = (cell[it] / cells) % cells, face_all = cell[it] / (cells * cells);
        const float *face = cubes + (size_t)face_all * res * res;
        auto take = [&](float stored) {
            const float closest = LAYOUT == VKR_CUBE_LAYOUT_TILED_8X4_X25 ? stored : stored * 25.0f;
            nan[it] |= closest != closest;
            lo[it] = __builtin_fminf(lo[it], closest); hi[it] = __builtin_fmaxf(hi[it], closest);
        };
#pragma unroll
        for (unsigned r = 0; r < 4; r++) {
            if (FAST) {
                size_t at;  // in floats
                if (LAYOUT == VKR_CUBE_LAYOUT_LINEAR) at = (size_t)(cy * RANGE_CELL + (q >> 2) + 4u * r) * res + cx * RANGE_CELL + (q & 3u) * 4u;
                else at = (size_t)(cx * 2u + (r >> 1)) * 8u * res + (size_t)cy * (RANGE_CELL * 8u) + ((r & 1u) * 16u + q) * 4u;
                typedef float float4_t __attribute__((ext_vector_type(4)));
                const float4_t v = __builtin_nontemporal_load((const float4_t *)(face + at));  // read once: no reason to keep it in a cache
                take(v.x); take(v.y); take(v.z); take(v.w);
            } else {
                const unsigned i = cx * RANGE_CELL + q;
#pragma unroll
                for (unsigned t = 0; t < 4; t++) {
                    const unsigned j = cy * RANGE_CELL + r * 4u + t;
                    if (i < (unsigned)res && j < (unsigned)res)
                        take(face[LAYOUT == VKR_CUBE_LAYOUT_LINEAR ? (size_t)j * res + i : (size_t)(i >> 3) * 8u * res + 8u * j + (i & 7u)]);
                }
            }
        }
    }
#pragma unroll
    for (unsigned it = 0; it < 4; it++) {
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {  // over the sixteen lanes of the cell
            lo[it] = __builtin_fminf(lo[it], __shfl_xor(lo[it], d)); hi[it] = __builtin_fmaxf(hi[it], __shfl_xor(hi[it], d));
        }
        // fmin / fmax drop a NaN: the cell is poisoned when any of its sixteen lanes saw one
        if ((__builtin_amdgcn_ballot_w64(nan[it]) >> (lane & 48u)) & 0xFFFFull) { lo[it] = -__builtin_inff(); hi[it] = __builtin_inff(); }
    }
#pragma unroll
    for (unsigned it = 0; it < 4; it++)
        if (q == 0 && cell[it] < total_cells) {
            const unsigned per_face = cells * cells, face_all = cell[it] / per_face, light = face_all / 6u, f = face_all - light * 6u;
            ranges[(((size_t)light * (RANGE_LEVELS + 1) + RANGE_LEVELS) * 6u + f) * per_face + (cell[it] - face_all * per_face)] = make_float2(lo[it], hi[it]);
        }
}

// Step 2, once per level in rising order: block (bx, by) of level 0 is the four cells (bx..bx+1, by..by+1); block
// (bx, by) of level k > 0 is the blocks (2bx, 2by), (2bx+2, 2by), (2bx, 2by+2), (2bx+2, 2by+2) of level k - 1, which
// tile it exactly; parts beyond the face's edge drop out.  One thread per entry of the padded level slab; entries
// past the last block of a row or column hold (-inf, +inf) and are never read.  A poisoned part is (-inf, +inf)
// and poisons the block.
__global__ __launch_bounds__(256) void cube_range_level_kernel(float2 *ranges, unsigned cells, unsigned n_faces, unsigned level)
{
    const unsigned per_face = cells * cells;
    const size_t id = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (id >= (size_t)n_faces * per_face) return;
    const unsigned at = (unsigned)(id % per_face), bx = at % cells, by = at / cells, face_all = (unsigned)(id / per_face);
    const unsigned light = face_all / 6u, f = face_all - light * 6u;
    float2 *of_light = ranges + (size_t)light * (RANGE_LEVELS + 1) * 6u * per_face;
    const float2 *part = of_light + ((size_t)(level == 0 ? RANGE_LEVELS : level - 1u) * 6u + f) * per_face;
    const unsigned blocks = (cells + (1u << level) - 1u) >> level;
    const unsigned parts = level == 0 ? cells : (cells + (1u << (level - 1u)) - 1u) >> (level - 1u);  // per row of the finer level
    const unsigned step = level == 0 ? 1u : 2u, x0 = level == 0 ? bx : 2u * bx, y0 = level == 0 ? by : 2u * by;
    float lo = -__builtin_inff(), hi = __builtin_inff();
    if (bx < blocks && by < blocks) {
        lo = __builtin_inff(); hi = -__builtin_inff();
#pragma unroll
        for (unsigned dy = 0; dy < 2; dy++)
#pragma unroll
            for (unsigned dx = 0; dx < 2; dx++) {
                const unsigned x = x0 + dx * step, y = y0 + dy * step;
                if (x < parts && y < parts) {
                    const float2 c = part[(size_t)y * cells + x];
                    lo = __builtin_fminf(lo, c.x); hi = __builtin_fmaxf(hi, c.y);
                }
            }
    }
    of_light[((size_t)level * 6u + f) * per_face + at] = make_float2(lo, hi);
}

// The largest value over the 64 lanes of the wave (every lane active, no NaN), as a wave-uniform value, without going
// through LDS: four data-parallel-primitive steps leave each row of 16 lanes holding its row's maximum (swap with the
// neighbour, with the other pair of the quad, with the mirrored half row, with the mirrored row), two broadcast steps
// carry the maxima across the rows into the last one, and lane 63 is read as a scalar.  Written as instructions: the
// compiler's form of a max is preceded by a canonicalising max of its own and a DPP move, three times the count.
// (A DPP operand written by the instruction before needs two wait states, which inline code has to provide itself.)
VKR_DEV float wave_max_uniform(float v)
{
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// What the depth ranges say about a whole 8x8-pixel tile and a light, decided once per pair instead of once per pixel:
// bit l of the first mask = every tap of every pixel of the tile that takes part in the light loop is in shadow for
// light l (the light adds exactly zero and the shading kernel skips it before computing anything), bit l of the second
// = none is (the kernel takes count 0 without looking).  One wave per tile.  Its lanes first read the tile's pixels
// and reduce the bounding box of their positions and their largest disk radius; then lane l bounds, for light l,
// what the shading kernel computes per pixel, using that every step is a correctly rounded, monotone operation:
//   fragToLight = RN(pos - lp) lies between the values at the box's ends, component by component; its squared
//   length (dot3: a product and two fmas of non-negative terms) between dot3 of the smallest and of the largest
//   magnitudes, so cur = RN(RN(sqrt) - 0.15) between cur_min and cur_max computed by the same operations;
//   a tap coordinate RN(f +- disk) between RN(f_lo - disk_max) and RN(f_hi + disk_max);
//   when the smallest magnitude of one coordinate's candidates exceeds the largest of both others, every tap of
//   every pixel selects that face (ties cannot occur) and the sign of the major coordinate is fixed;
//   a face coordinate is then +-u / |m| with u and |m| in known intervals: its extremes lie at the intervals' ends.
//   Texel indices are floors of t = q res / 2 + res / 2 <= 2^13, which the shading kernel evaluates from a correctly
//   rounded reciprocal, a product and an fma (or an fma and a product): within 2^-9 of the exact value.  The ends of
//   the interval are evaluated here from hardware reciprocals (one unit in the last place off), a product and an fma:
//   within 2^-9 as well.  So every tap's t lies within 2^-8 of [t_lo, t_hi] as computed here, and its floor between
//   floor(t_lo - 2^-7) and floor(t_hi + 2^-7).
// A window narrower than S = 16 << level lies inside one block of that level (see RANGE_CELL); its range
// against cur_min / cur_max decides.  Tiles with a non-finite position or disk radius among their pixels, distances
// outside the range where the kernel's own skip is valid (2^-96 <= |fragToLight|^2 <= 2^96), or windows too wide,
// get neither bit, and the shading kernel decides per pixel as before.
__global__ __launch_bounds__(256) void light_classes_kernel(const ShadeParams p, unsigned long long *classes)
{
    const unsigned tile = blockIdx.x * 4u + (threadIdx.x >> 6);  // four waves, a tile each
    if (tile >= (unsigned)(p.tiles_x * p.tiles_y)) return;       // a whole wave: the cross-lane steps below see 64 lanes or none
    const int tile_x = (int)(tile % (unsigned)p.tiles_x), tile_y = (int)(tile / (unsigned)p.tiles_x);
    const int lane = threadIdx.x & 63;
    const int x = tile_x * 8 + (lane & 7), y = tile_y * 8 + (lane >> 3);
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, disk_max = 0.0f;
    bool takes_part = false, bad = false;
    if (x < p.width && y < p.rows) {
        const size_t px = (size_t)y * p.width + x;
        const uint2 normal_bits = p.normal[px], position_bits = p.position[px];  // both loads in flight together
        const rgba no = unpack4(normal_bits);
        const f3 nraw = mk3(no.r, no.g, no.b);
        takes_part = dot3(nraw, nraw) > 0.0f;  // the shading kernel's own test
        if (takes_part) {
            const rgba po = unpack4(position_bits);
            const f3 pos = mk3(po.r, po.g, po.b);
            const f3 Vraw = sub3(mk3(p.cam[0], p.cam[1], p.cam[2]), pos);
            // an upper bound of the pixel's disk radius is all that is needed: the hardware root, one unit in the
            // last place off at most, with a margin, through the kernel's own (monotone) fma and product
            const float view_distance = __builtin_amdgcn_sqrtf(dot3(Vraw, Vraw)) * (1.0f + 0x1p-20f);
            const float disk = fma_(view_distance, 1.0f / 25.0f, 1.0f) * (1.0f / 25.0f);
            bad = !(__builtin_fabsf(pos.x) < inf && __builtin_fabsf(pos.y) < inf && __builtin_fabsf(pos.z) < inf && disk < inf);
            lo[0] = hi[0] = pos.x; lo[1] = hi[1] = pos.y; lo[2] = hi[2] = pos.z;
            disk_max = disk;
        }
    }
    const bool usable = __builtin_amdgcn_ballot_w64(takes_part) != 0ull && __builtin_amdgcn_ballot_w64(bad) == 0ull;
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = -wave_max_uniform(-lo[c]); hi[c] = wave_max_uniform(hi[c]); }
    disk_max = wave_max_uniform(disk_max);
    const float fres = (float)p.cube_res, half = 0.5f * fres;
    for (int word = 0; word < p.class_words; word++) {
        const int li = word * 64 + lane;
        bool skip = false, lit = false;
        if (usable && li < p.n_lights) {
            const float *lf = p.lights + (size_t)li * 23;
            const float lp[3] = {lf[16], lf[17], lf[18]};
            // an interval [a, b] with a <= b: its smallest magnitude is max(a, -b, 0), its largest max(-a, b)
            float f_lo[3], f_hi[3], a_min[3], a_max[3], c_lo[3], c_hi[3], c_min[3], c_max[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                f_lo[c] = lo[c] - lp[c]; f_hi[c] = hi[c] - lp[c];
                a_max[c] = __builtin_fmaxf(-f_lo[c], f_hi[c]);
                a_min[c] = __builtin_fmaxf(__builtin_fmaxf(f_lo[c], -f_hi[c]), 0.0f);
                c_lo[c] = f_lo[c] - disk_max; c_hi[c] = f_hi[c] + disk_max;
                c_max[c] = __builtin_fmaxf(-c_lo[c], c_hi[c]);
                c_min[c] = __builtin_fmaxf(__builtin_fmaxf(c_lo[c], -c_hi[c]), 0.0f);
            }
            const float d2_min = dot3(mk3(a_min[0], a_min[1], a_min[2]), mk3(a_min[0], a_min[1], a_min[2]));
            const float d2_max = dot3(mk3(a_max[0], a_max[1], a_max[2]), mk3(a_max[0], a_max[1], a_max[2]));
            const bool in_domain = d2_min >= 0x1p-96f && d2_max <= 0x1p96f;  // false for NaN
            // both roots are correctly rounded inside the domain just tested, and unused outside it
            const float cur_min = fast_sqrt_rn(d2_min) - 0.15f, cur_max = fast_sqrt_rn(d2_max) - 0.15f;
            // The faces a tap can select: major coordinate m with sign s needs a candidate of that sign whose magnitude
            // reaches the smallest magnitudes of both other coordinates.  Bit 2m + (s < 0).
            unsigned possible = 0u;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const float need = __builtin_fmaxf(c_min[(m + 1) % 3], c_min[(m + 2) % 3]);
                if (c_hi[m] > 0.0f && c_hi[m] >= need) possible |= 1u << (2 * m);
                if (c_lo[m] < 0.0f && -c_lo[m] >= need) possible |= 2u << (2 * m);
            }
            bool decided = in_domain && possible != 0u;  // until a face says otherwise
            float nearest = inf, farthest = -inf;        // over the ranges of every face a tap can fall on
            int faces_seen = 0, one_face = 0, one_i0 = 0, one_i1 = 0, one_j0 = 0, one_j1 = 0;  // the window of a tile that has one face only
            unsigned one_level = 0u;
            // One face in most tiles, two along a cube edge, three at a corner: more (a light inside the tile's box) are left
            // to the per-pixel path.  The lanes of a wave are different lights, so the trip count is the wave's largest.
            for (int trip = 0; trip < 3 && wave_any(decided && possible != 0u); trip++) {
                if (decided && possible != 0u) {
                    const int face = __builtin_ctz(possible);
                    possible &= possible - 1u;
                    const int m = face >> 1;
                    const bool negative = face & 1;
                    const float side_lo = m == 2 ? c_lo[2] : (m == 1 ? c_lo[1] : c_lo[0]), side_hi = m == 2 ? c_hi[2] : (m == 1 ? c_hi[1] : c_hi[0]);
                    // |major| over the taps of this face: on this side of zero, and at least every other coordinate's magnitude
                    const float m_max = negative ? -side_lo : side_hi;
                    const float others = __builtin_fmaxf(m == 0 ? c_min[1] : c_min[0], m == 2 ? c_min[1] : c_min[2]);
                    const float m_min = __builtin_fmaxf(__builtin_fmaxf(negative ? -side_hi : side_lo, others), 0.0f);
                    if (!(m_min > 0.0f && m_max < 0x1p100f)) { decided = false; continue; }
                    // sc and tc: which coordinate, and the sign it is taken with (taps_one_face)
                    int su_axis, sw_axis; float su_sign, sw_sign;
                    const float sgn = negative ? -1.0f : 1.0f;
                    if (m == 2) { su_axis = 0; su_sign = sgn; sw_axis = 1; sw_sign = -1.0f; }
                    else if (m == 1) { su_axis = 0; su_sign = 1.0f; sw_axis = 2; sw_sign = sgn; }
                    else { su_axis = 2; su_sign = -sgn; sw_axis = 1; sw_sign = -1.0f; }
                    const float r_small = __builtin_amdgcn_rcpf(m_max), r_large = __builtin_amdgcn_rcpf(m_min);  // one unit in the last place off at most
                    auto window = [&](int axis, float sign, int &first, int &last) {
                        float u_lo = axis == 0 ? c_lo[0] : (axis == 1 ? c_lo[1] : c_lo[2]);
                        float u_hi = axis == 0 ? c_hi[0] : (axis == 1 ? c_hi[1] : c_hi[2]);
                        if (sign < 0.0f) { const float t = u_lo; u_lo = -u_hi; u_hi = -t; }
                        // a tap on this face has |u| <= |major|: beyond +-1 (plus a rounding, absorbed by the margin below) lie only
                        // quotients of candidates that select another face
                        const float q_lo = __builtin_fmaxf(__builtin_fminf(u_lo * r_small, u_lo * r_large), -1.0f);
                        const float q_hi = __builtin_fminf(__builtin_fmaxf(u_hi * r_small, u_hi * r_large), 1.0f);
                        const float t_lo = fma_(q_lo, half, half), t_hi = fma_(q_hi, half, half);
                        first = max((int)__builtin_floorf(t_lo - 0x1p-7f), 0);
                        last = min((int)__builtin_floorf(t_hi + 0x1p-7f), p.cube_res - 1);
                    };
                    int i0, i1, j0, j1;
                    window(su_axis, su_sign, i0, i1);
                    window(sw_axis, sw_sign, j0, j1);
                    const unsigned width = (unsigned)max(max(i1 - i0, j1 - j0), 0);
                    unsigned level = 32u - (unsigned)__clz((int)(width >> 4));  // smallest k with width < 16 << k: its block always holds the window
                    // by where it starts, the window may also lie inside the half-sized block of the level below: columns
                    // [bx S', bx S' + 2 S') with S' = S / 2, bx = i0 / S' (no deeper: the window is at least S' wide)
                    if (level >= 1u && level <= (unsigned)RANGE_LEVELS) {
                        const unsigned sh = 3u + level;
                        if (((((unsigned)i0 >> sh) + 2u) << sh) > (unsigned)i1 && ((((unsigned)j0 >> sh) + 2u) << sh) > (unsigned)j1) level -= 1u;
                    }
                    if (level >= (unsigned)RANGE_LEVELS) { decided = false; continue; }
                    const unsigned shift = 4u + level;
                    const size_t per_face = (size_t)p.range_cells * p.range_cells;
                    const float2 range = p.cube_ranges[(((size_t)li * (RANGE_LEVELS + 1) + level) * 6u + (unsigned)face) * per_face +
                                                       (size_t)((unsigned)j0 >> shift) * p.range_cells + ((unsigned)i0 >> shift)];
                    nearest = __builtin_fminf(nearest, range.x);
                    farthest = __builtin_fmaxf(farthest, range.y);
                    faces_seen++; one_face = face; one_i0 = i0; one_i1 = i1; one_j0 = j0; one_j1 = j1; one_level = level;
                }
            }
            decided = decided && possible == 0u;
            skip = decided && cur_min > farthest;
            lit = decided && !(cur_max > nearest);
            // A second look for a one-face tile that a coarse block left undecided: the blocks of level 0 (32 x 32 texels)
            // at every other cell cover the cells of the window, up to three by three of them, more closely than one block of
            // a coarser level does (a window 40 texels wide sat in a block of 128).
            const unsigned cx0 = (unsigned)one_i0 >> 4, cy0 = (unsigned)one_j0 >> 4;
            const unsigned nx = ((((unsigned)one_i1 >> 4) - cx0) >> 1) + 1u, ny = ((((unsigned)one_j1 >> 4) - cy0) >> 1) + 1u;
            const bool again = decided && !skip && !lit && faces_seen == 1 && one_level >= 1u && nx <= 3u && ny <= 3u;
            if (wave_any(again)) {
                float lo2 = inf, hi2 = -inf;
                const float2 *level0 = p.cube_ranges + (((size_t)li * (RANGE_LEVELS + 1)) * 6u + (unsigned)one_face) * ((size_t)p.range_cells * p.range_cells);
#pragma unroll
                for (unsigned dy = 0; dy < 3; dy++)
#pragma unroll
                    for (unsigned dx = 0; dx < 3; dx++)
                        if (again && dx < nx && dy < ny) {
                            const float2 range = level0[(size_t)(cy0 + 2u * dy) * p.range_cells + (cx0 + 2u * dx)];
                            lo2 = __builtin_fminf(lo2, range.x);
                            hi2 = __builtin_fmaxf(hi2, range.y);
                        }
                if (again) { skip = cur_min > hi2; lit = !(cur_max > lo2); }
            }
        }
        const unsigned long long skip_mask = __builtin_amdgcn_ballot_w64(skip), lit_mask = __builtin_amdgcn_ballot_w64(lit);
        if (lane == 0) {
            classes[((size_t)tile * p.class_words + word) * 2] = skip_mask;
            classes[((size_t)tile * p.class_words + word) * 2 + 1] = lit_mask;
        }
    }
}

}  // namespace vkr

#ifdef VKR_EXPERIMENT_COUNT
extern "C" __attribute__((visibility("default"))) int32_t vkr_experiment_shade_counts(unsigned long long *out8, int32_t reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(vkr::shade_counts), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vkr::shade_counts), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int64_t vkr_cube_shadow_ranges_bytes(int32_t n_lights, int32_t res)
{
    using namespace vkr;
    if (n_lights < 0 || res <= 0 || res > 8192) return -1;
    const int64_t cells = (res + RANGE_CELL - 1) / RANGE_CELL;
    return (int64_t)n_lights * (RANGE_LEVELS + 1) * 6 * cells * cells * (int64_t)sizeof(float2);
}

extern "C" int32_t vkr_cube_shadow_ranges(const float *cubes, int32_t n_lights, int32_t res, int32_t layout, void *ranges, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(n_lights >= 0 && res > 0 && res <= 8192, "vkr_cube_shadow_ranges: bad size (%d lights, faces of %d)", n_lights, res);
    VKR_REQUIRE(layout >= VKR_CUBE_LAYOUT_LINEAR && layout <= VKR_CUBE_LAYOUT_TILED_8X4_X25, "vkr_cube_shadow_ranges: unknown cube shadow layout %d", layout);
    VKR_REQUIRE(layout == VKR_CUBE_LAYOUT_LINEAR || res % 8 == 0, "vkr_cube_shadow_ranges: the tiled layout needs a face size that is a multiple of 8 (got %d)", res);
    if (n_lights == 0) return VKR_OK;
    VKR_REQUIRE(cubes && ranges, "vkr_cube_shadow_ranges: a buffer is null");
    const unsigned cells = (unsigned)((res + RANGE_CELL - 1) / RANGE_CELL), n_faces = (unsigned)n_lights * 6u;
    VKR_REQUIRE((uint64_t)n_faces * cells * cells < (1ull << 31), "vkr_cube_shadow_ranges: too many cells for one launch (%d lights, faces of %d)", n_lights, res);
    const unsigned total_cells = n_faces * cells * cells;  // at most 65535 * 512^2 would overflow: bounded below
    typedef void (*kernel_t)(const float *, float2 *, int, unsigned, unsigned);
    static const kernel_t kernels[6] = {
        cube_range_cells_kernel<VKR_CUBE_LAYOUT_LINEAR, false>, cube_range_cells_kernel<VKR_CUBE_LAYOUT_LINEAR, true>,
        cube_range_cells_kernel<VKR_CUBE_LAYOUT_TILED_8X4, false>, cube_range_cells_kernel<VKR_CUBE_LAYOUT_TILED_8X4, true>,
        cube_range_cells_kernel<VKR_CUBE_LAYOUT_TILED_8X4_X25, false>, cube_range_cells_kernel<VKR_CUBE_LAYOUT_TILED_8X4_X25, true>};
    const bool fast = res % RANGE_CELL == 0 && ((uintptr_t)cubes & 15u) == 0;
    const unsigned per_block = 4u * RANGE_CELLS_PER_WAVE;
    hipLaunchKernelGGL(kernels[layout * 2 + (fast ? 1 : 0)], dim3((total_cells + per_block - 1u) / per_block), dim3(256), 0, as_stream(stream),
                       cubes, (float2 *)ranges, res, cells, total_cells);
    for (unsigned level = 0; level < (unsigned)RANGE_LEVELS; level++)
        hipLaunchKernelGGL(cube_range_level_kernel, dim3((total_cells + 255u) / 256u), dim3(256), 0, as_stream(stream), (float2 *)ranges, cells, n_faces, level);
    return launch_status("vkr_cube_shadow_ranges");
}

extern "C" int64_t vkr_shade_workspace_bytes(int32_t width, int32_t rows, int32_t n_lights)
{
    if (width <= 0 || rows <= 0 || n_lights < 0) return -1;
    const int64_t tiles = (int64_t)((width + 7) / 8) * ((rows + 7) / 8), words = (n_lights + 63) / 64;
    return tiles * words * 2 * (int64_t)sizeof(unsigned long long);
}

extern "C" int32_t vkr_shade(const VkrShadeArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_shade: args is null");
    VKR_REQUIRE(a->width > 0 && a->rows > 0, "vkr_shade: bad size %dx%d", a->width, a->rows);
    VKR_REQUIRE(a->albedo && a->normal && a->arm && a->position, "vkr_shade: a G-buffer image is null");
    VKR_REQUIRE(a->out_color && a->out_threshold, "vkr_shade: an output image is null");
    VKR_REQUIRE(a->n_lights >= 0, "vkr_shade: n_lights < 0");
    VKR_REQUIRE(a->n_lights == 0 || (a->lights && a->cube_shadow && a->cube_res > 0 && a->cube_res <= 8192), "vkr_shade: lights given without light data / cube shadow maps (face size 1..8192)");
    VKR_REQUIRE(a->dir_shadow && a->dir_res > 0, "vkr_shade: directional shadow map missing");
    VKR_REQUIRE(a->irradiance && a->irr_res > 0 && a->prefilter && a->pref_res > 0 && a->brdf_lut && a->lut_res > 0, "vkr_shade: an IBL image is missing");
    VKR_REQUIRE((int64_t)a->dir_res * a->dir_res < (1ll << 31), "vkr_shade: directional shadow map too large");
    VKR_REQUIRE(a->coverage_mode == VKR_COVERAGE_FULLSCREEN || a->coverage_mode == VKR_COVERAGE_MESH, "vkr_shade: unknown coverage mode %d", a->coverage_mode);
    VKR_REQUIRE(a->coverage_mode != VKR_COVERAGE_MESH || a->n_lights == 0 || a->coverage_masks, "vkr_shade: VKR_COVERAGE_MESH needs coverage_masks (see vkr_light_coverage)");
    VKR_REQUIRE(a->cube_layout >= VKR_CUBE_LAYOUT_LINEAR && a->cube_layout <= VKR_CUBE_LAYOUT_TILED_8X4_X25, "vkr_shade: unknown cube shadow layout %d", a->cube_layout);
    VKR_REQUIRE(a->cube_layout == VKR_CUBE_LAYOUT_LINEAR || a->n_lights == 0 || a->cube_res % 8 == 0, "vkr_shade: the tiled cube shadow layout needs a face size that is a multiple of 8 (got %d)", a->cube_res);

    ShadeParams p;
    p.width = a->width; p.rows = a->rows;
    p.albedo = (const uint2 *)a->albedo; p.normal = (const uint2 *)a->normal; p.arm = (const uint2 *)a->arm;
    p.position = (const uint2 *)a->position; p.background = (const uint2 *)a->background;
    p.lights = (const float *)a->lights; p.n_lights = a->n_lights;
    p.dir_shadow = a->dir_shadow; p.dir_res = a->dir_res;
    p.cube_shadow = a->cube_shadow; p.cube_res = a->cube_res;
    p.irradiance = a->irradiance; p.irr_res = a->irr_res;
    p.prefilter = a->prefilter; p.pref_res = a->pref_res;
    p.brdf_lut = (const uint32_t *)a->brdf_lut; p.lut_res = a->lut_res;
    p.out_color = (uint2 *)a->out_color; p.out_threshold = (uint2 *)a->out_threshold;
    p.coverage = a->coverage_mode == VKR_COVERAGE_MESH ? (const unsigned long long *)a->coverage_masks : nullptr;
    p.coverage_flags = p.coverage ? a->coverage_flags : nullptr;
    p.cube_ranges = a->n_lights > 0 ? (const float2 *)a->cube_ranges : nullptr;
    p.range_cells = (unsigned)((a->cube_res + RANGE_CELL - 1) / RANGE_CELL);
    p.class_words = (a->n_lights + 63) / 64;
    p.light_classes = nullptr;
    if (p.cube_ranges && a->workspace) {
        VKR_REQUIRE(a->workspace_bytes >= vkr_shade_workspace_bytes(a->width, a->rows, a->n_lights),
                    "vkr_shade: workspace of %lld bytes, vkr_shade_workspace_bytes() asks for %lld", (long long)a->workspace_bytes,
                    (long long)vkr_shade_workspace_bytes(a->width, a->rows, a->n_lights));
        p.light_classes = (const unsigned long long *)a->workspace;
    }
    for (int i = 0; i < 3; i++) {
        p.cam[i] = a->scene.camera_position[i];
        p.dir_dir[i] = a->scene.dir_light_direction[i];
        p.dir_color[i] = a->scene.dir_light_color[i];
    }
    for (int i = 0; i < 16; i++) p.light_space[i] = a->scene.light_space_matrix[i];

    p.tiles_x = (a->width + 7) / 8; p.tiles_y = (a->rows + 7) / 8;
    VKR_REQUIRE((int64_t)p.tiles_x * p.tiles_y < (1ll << 31), "vkr_shade: strip too large");
    const unsigned round_tiles = 8u * SHADE_RUN;  // whole rounds of one run per XCD (see tile_of_block)
    const dim3 grid(((unsigned)(p.tiles_x * p.tiles_y) + round_tiles - 1) / round_tiles * round_tiles);
    const bool pow2 = a->n_lights > 0 && (a->cube_res & (a->cube_res - 1)) == 0;
    const bool masked = p.coverage != nullptr;
    typedef void (*kernel_t)(const ShadeParams);
#define VKR_SHADE_FORMS(L, C) shade_kernel<L, false, false, C>, shade_kernel<L, false, true, C>, shade_kernel<L, true, false, C>, shade_kernel<L, true, true, C>
    static const kernel_t kernels[24] = {VKR_SHADE_FORMS(0, false), VKR_SHADE_FORMS(1, false), VKR_SHADE_FORMS(2, false),
                                         VKR_SHADE_FORMS(0, true), VKR_SHADE_FORMS(1, true), VKR_SHADE_FORMS(2, true)};
#undef VKR_SHADE_FORMS
    const kernel_t kernel = kernels[(p.light_classes ? 12 : 0) + a->cube_layout * 4 + (pow2 ? 2 : 0) + (masked ? 1 : 0)];
    // Many lights are drawn in several launches of `chunk` lights each: the waves of a launch drift apart as they
    // skip different lights, and with them the shadow cubes the resident waves of an XCD read from drift apart too,
    // until that XCD's L2 no longer holds them.  A cut costs a re-read of the G-buffer and the targets (48 bytes per
    // pixel) and changes no bit: between two draws the targets are binary16 values, in registers or in memory.
#ifdef VKR_EXPERIMENT_KNOBS  // kernel experiments only (make HIPFLAGS+=-DVKR_EXPERIMENT_KNOBS): the shipped library reads no environment
    static const int forced_chunk = getenv("VKR_SHADE_CHUNK") ? atoi(getenv("VKR_SHADE_CHUNK")) : -1;  // 0: one launch
    const int chunk = forced_chunk > 0 ? forced_chunk : (forced_chunk == 0 ? a->n_lights : SHADE_LIGHTS_PER_LAUNCH);
#else
    const int chunk = SHADE_LIGHTS_PER_LAUNCH;
#endif
    if (p.light_classes)
        hipLaunchKernelGGL(light_classes_kernel, dim3(((unsigned)(p.tiles_x * p.tiles_y) + 3u) / 4u), dim3(256), 0, as_stream(stream), p, (unsigned long long *)a->workspace);
    p.light_begin = 0;
    do {
        p.light_end = a->n_lights - p.light_begin <= chunk + chunk / 2 ? a->n_lights : p.light_begin + chunk;  // no short tail launch
        hipLaunchKernelGGL(kernel, grid, dim3(64), 0, as_stream(stream), p);
        p.light_begin = p.light_end;
    } while (p.light_begin < a->n_lights);
    return launch_status("vkr_shade");
}
