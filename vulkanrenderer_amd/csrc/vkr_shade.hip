// Composition pass on gfx950: the fullscreen composition.frag draw and every additive light.frag
// draw of VulkanRenderer.cs:445-498 in one kernel.  A pixel's G-buffer texels are read once and
// held in registers; the two RGBA16F render targets are accumulated in registers and rounded to
// binary16 after every light, exactly as the per-draw additive blend of
// GraphicsPipelineBuilder.cs:157-187 does, then written once.
//
// Work layout: a 16x16-pixel workgroup of four wave64s, each wave an 8x8-pixel block, so the 64
// lanes of a wave look up neighbouring shadow-map texels.  Light parameters are wave-uniform and
// come in through scalar loads.  The 20 cube-shadow taps of light.frag:131-137 only ever combine
// three x, three y and three z coordinates (v, v - r, v + r), so their face coordinates are
// computed once per coordinate pair instead of once per tap; when every tap of every lane of the
// wave falls on one cube face (the common case) the face selection disappears altogether.
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
    const float *dir_shadow; int dir_res;
    const float *cube_shadow; int cube_res;
    const void *irradiance; int irr_res;
    const void *prefilter; int pref_res;
    const uint32_t *brdf_lut; int lut_res;
    uint2 *out_color, *out_threshold;
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

// light.frag:20-27, as indices into {v - r, v, v + r}
#define VKR_GRID20(T)                                                                              \
    T(2, 2, 2) T(2, 0, 2) T(0, 0, 2) T(0, 2, 2) T(2, 2, 0) T(2, 0, 0) T(0, 0, 0) T(0, 2, 0)        \
    T(2, 2, 1) T(2, 0, 1) T(0, 0, 1) T(0, 2, 1) T(2, 1, 2) T(0, 1, 2) T(2, 1, 0) T(0, 1, 0)        \
    T(1, 2, 2) T(1, 0, 2) T(1, 0, 0) T(1, 2, 0)

// Texel coordinate (already clamped, as a byte offset when scale = 4) of face coordinate c / ma
VKR_DEV int face_texel(float c, float ima, float fres, int res)
{
    float s = fma_(c * ima, 0.5f, 0.5f);
    int i, r;
    // floor + convert in one instruction, clamp to [0, res-1] in one instruction
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(s * fres));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(res - 1));
    return r;
}

// Number of the 20 taps whose stored depth is closer than `cur` when every tap of every lane of
// the wave lies on cube face `face` (wave-uniform).  U, W are the coordinates that become sc, tc
// (sign already applied), IM the reciprocals of |major coordinate|; the index triple of a tap is
// permuted so that (iu, iw, im) index U, W, IM.
template <int AXIS>
VKR_DEV int taps_one_face(const float *face_base, int res, const float (&X)[3], const float (&Y)[3], const float (&Z)[3], bool neg, float cur)
{
    const float fres = (float)res;
    const float (&M)[3] = AXIS == 2 ? Z : (AXIS == 1 ? Y : X);
    float IM[3];
#pragma unroll
    for (int k = 0; k < 3; k++) IM[k] = fast_rcp_rn(__builtin_fabsf(M[k]));
    // z major: sc = +-x, tc = -y;  y major: sc = x, tc = +-z;  x major: sc = -+z, tc = -y
    int I[3][3], J[3][3];  // [index into U / W][index into M]
    const int res4 = res * 4;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int m = 0; m < 3; m++) {
            if (a == 1 && m == 1) { I[a][m] = 0; J[a][m] = 0; continue; }  // no tap uses (v, v)
            float u, w;
            if (AXIS == 2) { u = neg ? -X[a] : X[a]; w = -Y[a]; }
            else if (AXIS == 1) { u = X[a]; w = neg ? -Z[a] : Z[a]; }
            else { u = neg ? Z[a] : -Z[a]; w = -Y[a]; }
            I[a][m] = face_texel(u, IM[m], fres, res) * 4;
            J[a][m] = face_texel(w, IM[m], fres, res);
        }
    int count = 0;
    const char *base = (const char *)face_base;
#define VKR_TAP(gx, gy, gz)                                                              \
    {                                                                                    \
        constexpr int iu = AXIS == 0 ? gz : gx, iw = AXIS == 1 ? gz : gy, im = AXIS == 2 ? gz : (AXIS == 1 ? gy : gx); \
        unsigned off = __umul24((unsigned)J[iw][im], (unsigned)res4) + (unsigned)I[iu][im]; \
        float closest = *(const float *)(base + off) * 25.0f;                            \
        count += cur > closest ? 1 : 0;                                                  \
    }
    VKR_GRID20(VKR_TAP)
#undef VKR_TAP
    return count;
}

// The same count with the cube face chosen per tap and per lane (light.frag:133 through
// shadowSampler: NEAREST on the face the direction's major axis selects).
VKR_DEV int taps_any_face(const float *cube, int res, const float (&X)[3], const float (&Y)[3], const float (&Z)[3], float cur)
{
    const float fres = (float)res;
    float AX[3], AY[3], AZ[3], IX[3], IY[3], IZ[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        AX[k] = __builtin_fabsf(X[k]); AY[k] = __builtin_fabsf(Y[k]); AZ[k] = __builtin_fabsf(Z[k]);
        IX[k] = fast_rcp_rn(AX[k]); IY[k] = fast_rcp_rn(AY[k]); IZ[k] = fast_rcp_rn(AZ[k]);
    }
    const unsigned res4 = (unsigned)res * 4u;
    const char *base = (const char *)cube;
    int count = 0;
    // sign bits of the nine candidates: "c >= 0 ? v : -v" becomes v ^ sign(c) (they differ only for
    // c = -0, which can be the major coordinate only when the whole direction is zero)
    unsigned SX[3], SY[3], SZ[3], RX[3], RY[3], RZ[3];  // R*: first texel row of the face a candidate selects
#pragma unroll
    for (int k = 0; k < 3; k++) {
        SX[k] = __builtin_bit_cast(unsigned, X[k]) & 0x80000000u;
        SY[k] = __builtin_bit_cast(unsigned, Y[k]) & 0x80000000u;
        SZ[k] = __builtin_bit_cast(unsigned, Z[k]) & 0x80000000u;
        RX[k] = __umul24(SX[k] >> 31, (unsigned)res);
        RY[k] = __umul24(2u | (SY[k] >> 31), (unsigned)res);
        RZ[k] = __umul24(4u | (SZ[k] >> 31), (unsigned)res);
    }
#define VKR_TAP(gx, gy, gz)                                                                        \
    {                                                                                              \
        const float rx = X[gx], ry = Y[gy], rz = Z[gz], ax = AX[gx], ay = AY[gy], az = AZ[gz];     \
        const bool zm = (az >= ax) & (az >= ay);                                                   \
        const bool ym = !zm & (ay >= ax);                                                          \
        const float scz = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rx) ^ SZ[gz]);    \
        const float scx = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rz) ^ SX[gx] ^ 0x80000000u); \
        const float tcy = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rz) ^ SY[gy]);    \
        const float ima = zm ? IZ[gz] : (ym ? IY[gy] : IX[gx]);                                    \
        const float sc = zm ? scz : (ym ? rx : scx);                                               \
        const float tc = ym ? tcy : -ry;                                                           \
        const unsigned row0 = zm ? RZ[gz] : (ym ? RY[gy] : RX[gx]);                                \
        const unsigned i4 = (unsigned)face_texel(sc, ima, fres, res) * 4u;                         \
        const unsigned j = (unsigned)face_texel(tc, ima, fres, res);                               \
        const unsigned off = __umul24(row0 + j, res4) + i4;                                        \
        const float closest = *(const float *)(base + off) * 25.0f;                                \
        count += cur > closest ? 1 : 0;                                                            \
    }
    VKR_GRID20(VKR_TAP)
#undef VKR_TAP
    return count;
}

// Which single cube face all 27 combinations of {v-r, v, v+r}^3 select, or -1 if they do not
// agree.  The comparisons are those of the face-selection rule (ties: z over y over x) applied to
// the extreme candidates, so "agree" here implies the per-tap rule picks that face for each tap.
VKR_DEV int common_face(const float (&X)[3], const float (&Y)[3], const float (&Z)[3])
{
    float ax0 = __builtin_fabsf(X[0]), ax1 = __builtin_fabsf(X[1]), ax2 = __builtin_fabsf(X[2]);
    float ay0 = __builtin_fabsf(Y[0]), ay1 = __builtin_fabsf(Y[1]), ay2 = __builtin_fabsf(Y[2]);
    float az0 = __builtin_fabsf(Z[0]), az1 = __builtin_fabsf(Z[1]), az2 = __builtin_fabsf(Z[2]);
    float xmin = __builtin_fminf(ax0, __builtin_fminf(ax1, ax2)), xmax = __builtin_fmaxf(ax0, __builtin_fmaxf(ax1, ax2));
    float ymin = __builtin_fminf(ay0, __builtin_fminf(ay1, ay2)), ymax = __builtin_fmaxf(ay0, __builtin_fmaxf(ay1, ay2));
    float zmin = __builtin_fminf(az0, __builtin_fminf(az1, az2)), zmax = __builtin_fmaxf(az0, __builtin_fmaxf(az1, az2));
    int face = -1;
    if (zmin >= xmax && zmin >= ymax) face = 4;            // z major for every combination
    else if (zmax < ymin && ymin >= xmax) face = 2;        // never z major, always y over x
    else if (zmax < xmin && ymax < xmin) face = 0;         // never z major, never y over x
    if (face < 0) return -1;
    float m0 = face == 4 ? Z[0] : (face == 2 ? Y[0] : X[0]);
    float m1 = face == 4 ? Z[1] : (face == 2 ? Y[1] : X[1]);
    float m2 = face == 4 ? Z[2] : (face == 2 ? Y[2] : X[2]);
    bool pos = m0 >= 0.0f && m1 >= 0.0f && m2 >= 0.0f;
    bool neg = m0 < 0.0f && m1 < 0.0f && m2 < 0.0f;
    if (!pos && !neg) return -1;
    return face + (neg ? 1 : 0);
}

// Workgroups are dealt to the 8 XCDs round-robin by id, and each XCD has its own L2.  This maps
// block id -> tile so that each XCD shades one contiguous run of tiles, ordered in 16-tile-wide
// column blocks: the workgroups resident on an XCD at one time cover a compact screen region and
// therefore a compact region of every shadow map, which keeps their shared texels in that L2.
VKR_DEV void tile_of_block(unsigned b, int tiles_x, int tiles_y, int &tx, int &ty)
{
    const unsigned total = (unsigned)tiles_x * tiles_y;
    const unsigned xcd = b & 7u, q = b >> 3;
    // ids 0..total-1 dealt round-robin: XCD k owns ceil((total - k) / 8) of them
    const unsigned base = total >> 3, rem = total & 7u;
    const unsigned v = xcd * base + min(xcd, rem) + q;  // position in the column-block order
    const unsigned block_tiles = 16u * (unsigned)tiles_y;
    const unsigned cb = min(v / block_tiles, (unsigned)(tiles_x - 1) / 16u);
    const unsigned r = v - cb * block_tiles;
    const unsigned cbw = min(16u, (unsigned)tiles_x - cb * 16u);
    ty = (int)(r / cbw);
    tx = (int)(cb * 16u + r % cbw);
}

__global__ __launch_bounds__(256) void shade_kernel(const ShadeParams p)
{
    // 16x16 pixel tile per workgroup; wave w covers the 8x8 block (w & 1, w >> 1) of the tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile_x, tile_y;
    tile_of_block(blockIdx.x, p.tiles_x, p.tiles_y, tile_x, tile_y);
    const int x = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
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
    if (p.background) { rgba bg = unpack4(p.background[px]); c0 = bg.r; c1 = bg.g; c2 = bg.b; }

    if (al.a != 0.0f) {  // composition.frag:39-43
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
    c0 = round_f16(c0); c1 = round_f16(c1); c2 = round_f16(c2);
    t0 = round_f16(t0); t1 = round_f16(t1); t2 = round_f16(t2);

    // A cleared G-buffer texel has normal (0,0,0): the light shader's result for it is 0 under
    // IEEE maxNum semantics, so such pixels skip the light loop.
    if (!degenerate_normal) {
        const float disk = fma_(view_distance, 1.0f / 25.0f, 1.0f) * (1.0f / 25.0f);
        const float fourNdotV = 4.0f * NdotV;
        const size_t face_stride = (size_t)p.cube_res * p.cube_res;
        auto rcp_fast = [](float v) { return fast_rcp_rn(v); };
        for (int li = 0; li < p.n_lights; li++) {
            const float *lf = p.lights + li * 23;
            const f3 lp = mk3(lf[16], lf[17], lf[18]);
            const f3 lc = mk3(lf[20], lf[21], lf[22]);
            const f3 Lv = sub3(lp, pos);
            float dist, inv_dist;
            sqrt_and_rcp_rn(dot3(Lv, Lv), dist, inv_dist);
            const f3 L = scale3(Lv, inv_dist);
            const f3 Hv = add3(V, L);
            float hlen, inv_hlen;
            sqrt_and_rcp_rn(dot3(Hv, Hv), hlen, inv_hlen);
            const f3 H = scale3(Hv, inv_hlen);
            const float att = rcp_rn_checked(dist * dist);
            // the denominators below lie in [1e-4, 8]: well inside fast_rcp_rn's exact domain
            const float D = normal_distribution(dot3(N, H), a2, rcp_fast);
            const float NdotL = max0(dot3(N, L));
            const float G = g1V * geometry_schlick(NdotL, k, omk, rcp_fast);
            const f3 F = fresnel_schlick(max0(dot3(H, V)), F0, omF0);
            const float isd = fast_rcp_rn(fma_(fourNdotV, NdotL, 0.0001f));
            const float DG = D * G;

            // light.frag:121-141; |pos - lp| equals |lp - pos| bit for bit
            const float cur = dist - 0.15f;
            const float X[3] = {pos.x - lp.x - disk, pos.x - lp.x, pos.x - lp.x + disk};
            const float Y[3] = {pos.y - lp.y - disk, pos.y - lp.y, pos.y - lp.y + disk};
            const float Z[3] = {pos.z - lp.z - disk, pos.z - lp.z, pos.z - lp.z + disk};
            const float *cube = p.cube_shadow + (size_t)li * 6 * face_stride;
            const int face = common_face(X, Y, Z);
            const int face0 = __builtin_amdgcn_readfirstlane(face);
            int count;
            if (face0 >= 0 && __all(face == face0)) {
                const float *fb = cube + (size_t)face0 * face_stride;
                const bool neg = face0 & 1;
                if (face0 >= 4) count = taps_one_face<2>(fb, p.cube_res, X, Y, Z, neg, cur);
                else if (face0 >= 2) count = taps_one_face<1>(fb, p.cube_res, X, Y, Z, neg, cur);
                else count = taps_one_face<0>(fb, p.cube_res, X, Y, Z, neg, cur);
            } else {
                count = taps_any_face(cube, p.cube_res, X, Y, Z, cur);
            }
            const float lit = 1.0f - (float)count * (1.0f / 20.0f);

            const float o0 = ((fma_((1.0f - F.x) * albedo.x, INV_PI, (DG * F.x) * isd) * (lc.x * att)) * lit) * NdotL;
            const float o1 = ((fma_((1.0f - F.y) * albedo.y, INV_PI, (DG * F.y) * isd) * (lc.y * att)) * lit) * NdotL;
            const float o2 = ((fma_((1.0f - F.z) * albedo.z, INV_PI, (DG * F.z) * isd) * (lc.z * att)) * lit) * NdotL;
            const bool bright = luminance(o0, o1, o2) > 1.0f;  // light.frag:78-82
            c0 = round_f16(c0 + o0); c1 = round_f16(c1 + o1); c2 = round_f16(c2 + o2);
            t0 = round_f16(t0 + (bright ? o0 : 0.0f));
            t1 = round_f16(t1 + (bright ? o1 : 0.0f));
            t2 = round_f16(t2 + (bright ? o2 : 0.0f));
        }
    }
    p.out_color[px] = pack4(c0, c1, c2, 1.0f);
    p.out_threshold[px] = pack4(t0, t1, t2, 1.0f);
}

}  // namespace vkr

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
    if (a->coverage_mode != VKR_COVERAGE_FULLSCREEN)
        return fail(VKR_ERR_UNSUPPORTED, "vkr_shade: coverage mode %d is not implemented", a->coverage_mode);

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
    for (int i = 0; i < 3; i++) {
        p.cam[i] = a->scene.camera_position[i];
        p.dir_dir[i] = a->scene.dir_light_direction[i];
        p.dir_color[i] = a->scene.dir_light_color[i];
    }
    for (int i = 0; i < 16; i++) p.light_space[i] = a->scene.light_space_matrix[i];

    p.tiles_x = (a->width + 15) / 16; p.tiles_y = (a->rows + 15) / 16;
    VKR_REQUIRE((int64_t)p.tiles_x * p.tiles_y < (1ll << 31), "vkr_shade: strip too large");
    hipLaunchKernelGGL(shade_kernel, dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(256), 0, as_stream(stream), p);
    return launch_status("vkr_shade");
}
