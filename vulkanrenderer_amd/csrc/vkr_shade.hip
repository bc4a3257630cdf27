// Composition pass on gfx950: the fullscreen composition.frag draw and every additive light.frag
// draw of VulkanRenderer.cs:445-498 in one kernel.  A pixel's G-buffer texels are read once and
// held in registers; the two RGBA16F render targets are accumulated in registers and rounded to
// binary16 after every light, exactly as the per-draw additive blend of
// GraphicsPipelineBuilder.cs:157-187 does, then written once.
#include "vkr_device.h"
#include "vkr_host.h"

namespace vkr {

static constexpr float PI_F = 3.14159265359f;
static constexpr float INV_PI = 1.0f / PI_F;
static constexpr float LUM_R = 0.2126f, LUM_G = 0.7152f, LUM_B = 0.0722f;

struct ShadeParams {
    int width, rows;
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

// composition.frag:113-123
VKR_DEV float normal_distribution(float NdotH_raw, float a2)
{
    float NdotH = max0(NdotH_raw);
    float NdotH2 = NdotH * NdotH;
    float denom = fma_(NdotH2, a2 - 1.0f, 1.0f);
    denom = fma_(PI_F * denom, denom, 0.0001f);
    return a2 * rcp_rn(denom);
}

// composition.frag:125-130 with k = (roughness+1)^2/8 precomputed
VKR_DEV float geometry_schlick(float NdotX, float k) { return NdotX * rcp_rn(fma_(NdotX, 1.0f - k, k)); }

// composition.frag:141-143
VKR_DEV f3 fresnel_schlick(float cosTheta, f3 F0)
{
    float p = pow5(clamp01(1.0f - cosTheta));
    return mk3(fma_(1.0f - F0.x, p, F0.x), fma_(1.0f - F0.y, p, F0.y), fma_(1.0f - F0.z, p, F0.z));
}

VKR_DEV float luminance(f3 c) { return fma_(c.z, LUM_B, fma_(c.y, LUM_G, c.x * LUM_R)); }

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

// light.frag:20-27
__constant__ float GRID_DISK[20][3] = {
    {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, 1},
    {1, 1, -1}, {1, -1, -1}, {-1, -1, -1}, {-1, 1, -1},
    {1, 1, 0}, {1, -1, 0}, {-1, -1, 0}, {-1, 1, 0},
    {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1},
    {0, 1, 1}, {0, -1, 1}, {0, -1, -1}, {0, 1, -1}};

__global__ __launch_bounds__(256) void shade_kernel(const ShadeParams p)
{
    // 16x16 pixel tile per workgroup; each wave covers 16x4 pixels (128-byte row segments)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 16 + (lane & 15);
    const int y = blockIdx.y * 16 + wave * 4 + (lane >> 4);
    if (x >= p.width || y >= p.rows) return;
    const size_t px = (size_t)y * p.width + x;

    const rgba al = unpack4(p.albedo[px]);
    const rgba no = unpack4(p.normal[px]);
    const rgba ar = unpack4(p.arm[px]);
    const rgba po = unpack4(p.position[px]);
    const f3 albedo = mk3(al.r, al.g, al.b), nraw = mk3(no.r, no.g, no.b), pos = mk3(po.r, po.g, po.b);
    const float ao = ar.r, rough = ar.g, metal = ar.b;
    const f3 cam = mk3(p.cam[0], p.cam[1], p.cam[2]);

    // shared by both shaders: composition.frag:45-54, light.frag:43-51
    const float om = 1.0f - metal;
    const f3 F0 = mk3(fma_(albedo.x, metal, 0.04f * om), fma_(albedo.y, metal, 0.04f * om), fma_(albedo.z, metal, 0.04f * om));
    const f3 N = normalize3(nraw);
    const f3 Vraw = sub3(cam, pos);
    const float view_distance = sqrt_rn(dot3(Vraw, Vraw));
    const f3 V = scale3(Vraw, rcp_rn(view_distance));
    const float NdotV = max0(dot3(N, V));
    const float a = rough * rough, a2 = a * a;
    const float r1 = rough + 1.0f, k = (r1 * r1) * 0.125f;
    const float g1V = geometry_schlick(NdotV, k);
    const bool degenerate_normal = !(dot3(nraw, nraw) > 0.0f);

    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    if (p.background) { rgba bg = unpack4(p.background[px]); c0 = bg.r; c1 = bg.g; c2 = bg.b; }

    if (al.a != 0.0f) {  // composition.frag:39-43
        f3 Lo;
        {
            f3 L = normalize3(neg3(mk3(p.dir_dir[0], p.dir_dir[1], p.dir_dir[2])));
            f3 H = normalize3(add3(V, L));
            float D = normal_distribution(dot3(N, H), a2);
            float NdotL = max0(dot3(N, L));
            float G = g1V * geometry_schlick(NdotL, k);
            f3 F = fresnel_schlick(max0(dot3(H, V)), F0);
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
            f3 F = fresnel_schlick(NdotV, F0);
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
        bool bright = luminance(Lo) > 1.0f;
        t0 = bright ? Lo.x : 0.0f; t1 = bright ? Lo.y : 0.0f; t2 = bright ? Lo.z : 0.0f;
    }
    c0 = round_f16(c0); c1 = round_f16(c1); c2 = round_f16(c2);
    t0 = round_f16(t0); t1 = round_f16(t1); t2 = round_f16(t2);

    if (!degenerate_normal) {
        const float disk = fma_(view_distance, 1.0f / 25.0f, 1.0f) * (1.0f / 25.0f);
        const float fourNdotV = 4.0f * NdotV;
        const size_t cube_stride = (size_t)6 * p.cube_res * p.cube_res;
        for (int li = 0; li < p.n_lights; li++) {
            const float *lf = p.lights + li * 23;
            const f3 lp = mk3(lf[16], lf[17], lf[18]);
            const f3 lc = mk3(lf[20], lf[21], lf[22]);
            const f3 Lv = sub3(lp, pos);
            const float dist = sqrt_rn(dot3(Lv, Lv));
            const f3 L = scale3(Lv, rcp_rn(dist));
            const f3 H = normalize3(add3(V, L));
            const float att = rcp_rn(dist * dist);
            const float D = normal_distribution(dot3(N, H), a2);
            const float NdotL = max0(dot3(N, L));
            const float G = g1V * geometry_schlick(NdotL, k);
            const f3 F = fresnel_schlick(max0(dot3(H, V)), F0);
            const float isd = rcp_rn(fma_(fourNdotV, NdotL, 0.0001f));
            const float DG = D * G;

            // light.frag:121-141; |pos - lp| == |lp - pos| bit for bit
            const f3 l2f = sub3(pos, lp);
            const float *cube = p.cube_shadow + (size_t)li * cube_stride;
            const float cur = dist - 0.15f;
            float shadow = 0.0f;
#pragma unroll
            for (int i = 0; i < 20; i++) {
                f3 d = mk3(fma_(GRID_DISK[i][0], disk, l2f.x), fma_(GRID_DISK[i][1], disk, l2f.y), fma_(GRID_DISK[i][2], disk, l2f.z));
                float closest = texcube_depth_nearest(cube, p.cube_res, d) * 25.0f;
                shadow += cur > closest ? 1.0f : 0.0f;
            }
            const float lit = 1.0f - shadow * (1.0f / 20.0f);

            const float o0 = ((fma_((1.0f - F.x) * albedo.x, INV_PI, (DG * F.x) * isd) * (lc.x * att)) * lit) * NdotL;
            const float o1 = ((fma_((1.0f - F.y) * albedo.y, INV_PI, (DG * F.y) * isd) * (lc.y * att)) * lit) * NdotL;
            const float o2 = ((fma_((1.0f - F.z) * albedo.z, INV_PI, (DG * F.z) * isd) * (lc.z * att)) * lit) * NdotL;
            const bool bright = luminance(mk3(o0, o1, o2)) > 1.0f;  // light.frag:78-82
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
    VKR_REQUIRE(a->n_lights == 0 || (a->lights && a->cube_shadow && a->cube_res > 0), "vkr_shade: lights given without light data / cube shadow maps");
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

    dim3 grid((a->width + 15) / 16, (a->rows + 15) / 16);
    hipLaunchKernelGGL(shade_kernel, grid, dim3(256), 0, as_stream(stream), p);
    return launch_status("vkr_shade");
}
