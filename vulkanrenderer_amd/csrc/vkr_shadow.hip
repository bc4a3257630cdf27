// Shadow-map generation on gfx950: the point-light depth cubes of VulkanRenderer.cs:325-376
// (point_shadow.vert / .geom / .frag) and the directional depth map of VulkanRenderer.cs:406-428
// (depth.vert), both with FRONT faces culled, CCW front, depth test LESS on a target cleared to
// 1.0 (GraphicsPipeline.cs:262-264, 279-281).
//
// A texel belongs to a triangle when its centre lies inside the projected back-facing triangle
// and the depth there passes the 0 <= z <= w clip -- the homogeneous binary64 rule of the
// light-volume coverage kernel -- and the world position the fragment shader receives is the
// perspective-correct interpolation sum(Ei * Pi) * (1 / sum(Ei)) in binary32 (the edge values Ei rounded to it).  The stored value is the minimum over
// all covering triangles (atomicMin on the bits of a non-negative float), so the result does not
// depend on the order of triangles, lanes or launches.
//
// Work layout: one lane per (light, triangle); it projects the triangle onto the six faces in
// turn and walks the texels of its bounding box.  Output texels are addressed row-major or as
// 8x4-texel tiles, the layout the shading kernel gathers from fastest.
#include "vkr_device.h"
#include "vkr_fastmath.h"
#include "vkr_host.h"
#include "vkr_raster.h"

namespace vkr {

struct ShadowParams {
    const float *positions; const unsigned *indices; int n_verts, n_tris;
    float model[16];
    const float *lights;  // VkrLightInfo array viewed as floats (point) or null (directional)
    float light_space[16];  // directional: lightSpaceMatrix
    int res; float far_plane; int tiled; float store_scale;  // store_scale: 25 for VKR_CUBE_LAYOUT_TILED_8X4_X25, else 1
    float *out;
};

// CreateLookAt(eye, eye + dir, up) * CreatePerspectiveFieldOfView(pi/2, 1, 1, far), VulkanRenderer.cs:327-339.
// The look-at axes are exact signed unit vectors; 1/tan(pi/4) is taken as exactly 1.
__device__ const float SHADOW_DIR[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
__device__ const float SHADOW_UP[6][3] = {{0, -1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, -1, 0}, {0, -1, 0}};

VKR_DEV void shadow_face_matrix_elem(const float *eye, int face, float far_plane, int idx, float *out16)
{
    const float *d = SHADOW_DIR[face], *u = SHADOW_UP[face];
    const float z[3] = {-d[0], -d[1], -d[2]};
    const float x[3] = {u[1] * z[2] - u[2] * z[1], u[2] * z[0] - u[0] * z[2], u[0] * z[1] - u[1] * z[0]};
    const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    float view[16] = {x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0, 0, 0, 0, 1};
    view[12] = -fma_(x[2], eye[2], fma_(x[1], eye[1], x[0] * eye[0]));
    view[13] = -fma_(y[2], eye[2], fma_(y[1], eye[1], y[0] * eye[0]));
    view[14] = -fma_(z[2], eye[2], fma_(z[1], eye[1], z[0] * eye[0]));
    const float range = far_plane * rcp_rn(1.0f - far_plane);
    const float proj[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, range, -1, 0, 0, range, 0};
    out16[idx] = mat4_mul_elem(view, proj, idx >> 2, idx & 3);
}

VKR_DEV size_t depth_texel_index(int res, int i, int j, bool tiled)
{
    if (!tiled) return (size_t)j * res + i;
    return (size_t)((unsigned)i >> 3) * ((unsigned)res * 8u) + (unsigned)j * 8u + ((unsigned)i & 7u);  // tiles column by column
}

// The triangles of the wave's lanes into one res x res depth image each (`img` may differ per lane only
// through `have`: all lanes of a call target the same image).  POINT: world = corners' world positions
// and the depth is point_shadow.frag's; otherwise the fixed-function z / w.
template <bool POINT>
VKR_DEV void shadow_raster(RasterPool &pool, bool valid, const float (&clip)[3][4], const float (&world)[3][4], const float *light_pos, float inv_far, int res, bool tiled, float store_scale, float *img)
{
    int bx0 = 0, by0 = 0, bx1 = -1, by1 = -1;
    bool have = valid && !clip_reject(clip);
    if (have) {
        STri t;
        raster_setup(clip, t);
        have = t.D > 0.0 && shadow_bbox(clip, res, bx0, by0, bx1, by1);  // front faces and degenerate triangles are culled
    }
    const double sc = 2.0 / (double)res;
    raster_boxes<false, !POINT>(pool, have, clip, world, 0u, bx0, by0, bx1, by1, [&](const STri &t, const float (&wp)[3][4], unsigned, int i, int j) {
        const double xn = __builtin_fma((double)i + 0.5, sc, -1.0), yn = __builtin_fma((double)j + 0.5, sc, -1.0);
        const double e0 = s_edge(t, 0, xn, yn), e1 = s_edge(t, 1, xn, yn), e2 = s_edge(t, 2, xn, yn);
        const double s = (e0 + e1) + e2;
        if (!(e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0 && s > 0.0)) return;
        const double zs = __builtin_fma(e2, t.z[2], __builtin_fma(e1, t.z[1], e0 * t.z[0]));
        if (!(zs >= 0.0 && zs <= t.D)) return;
        float depth;
        if (POINT) {  // point_shadow.frag:11-16 on the interpolated world position
            // coverage was decided in binary64; the attribute is interpolated as graphics hardware does it, in binary32:
            // the edge values rounded to binary32 are the perspective-correct weights
            float pw[3];
            const float f0 = (float)e0, f1 = (float)e1, f2 = (float)e2;
            const float inv_s = rcp_rn_checked((f0 + f1) + f2);
#pragma unroll
            for (int k = 0; k < 3; k++)
                pw[k] = fma_(f2, wp[2][k], fma_(f1, wp[1][k], f0 * wp[0][k])) * inv_s;
            const f3 d = mk3(pw[0] - light_pos[0], pw[1] - light_pos[1], pw[2] - light_pos[2]);
            depth = sqrt_rn_checked(dot3(d, d)) * inv_far;
        } else {
            depth = (float)(zs * t.inv_D) + 0.0f;  // z / w of the fixed-function pipeline as z * (1/w) (-0 becomes +0)
        }
        // CompareOp.Less against the cleared 1.0 and every earlier write; depth >= 0 here.  The stored value may be
        // the product with 25 the shading pass forms: rounding is monotonic, so the minimum of the products is the
        // product of the minimum.
        if (depth < 1.0f)
            atomicMin((unsigned int *)(img + depth_texel_index(res, i, j, tiled)), __builtin_bit_cast(unsigned int, depth * store_scale));
    });
}

#ifndef VKR_PS_BLOCK
#define VKR_PS_BLOCK 64
#endif
__global__ __launch_bounds__(VKR_PS_BLOCK) void point_shadow_kernel(const ShadowParams p)
{
    __shared__ float s_m[6][16];
    __shared__ RasterPool pools[VKR_PS_BLOCK / 64];
    RasterPool &pool = pools[threadIdx.x >> 6];
    const int li = blockIdx.y;
    const float *lp = p.lights + (size_t)li * 23 + 16;
    const float eye[3] = {lp[0], lp[1], lp[2]};
    for (int e = threadIdx.x; e < 96; e += VKR_PS_BLOCK) shadow_face_matrix_elem(eye, e >> 4, p.far_plane, e & 15, s_m[e >> 4]);
    __syncthreads();
    const int t = blockIdx.x * VKR_PS_BLOCK + threadIdx.x;
    const bool valid = t < p.n_tris;  // lanes past the end stay: the wave shares large triangles
    float world[3][4] = {};
    if (valid) {
        const unsigned vmax = (unsigned)p.n_verts - 1u;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float *v = p.positions + 3 * (size_t)min(p.indices[3 * t + k], vmax);
            const float v4[4] = {v[0], v[1], v[2], 1.0f};
            mat4_vec4(p.model, v4, world[k]);  // point_shadow.vert:12
        }
    }
    const float inv_far = rcp_rn(p.far_plane);
    const size_t face_texels = (size_t)p.res * p.res;
    for (int face = 0; face < 6; face++) {
        float clip[3][4];
#pragma unroll
        for (int k = 0; k < 3; k++) mat4_vec4(s_m[face], world[k], clip[k]);  // point_shadow.geom:20
        shadow_raster<true>(pool, valid, clip, world, eye, inv_far, p.res, p.tiled != 0, p.store_scale, p.out + ((size_t)li * 6 + face) * face_texels);
    }
}

__global__ __launch_bounds__(256) void dir_shadow_kernel(const ShadowParams p)
{
    __shared__ float s_mvp[16];
    __shared__ RasterPool pools[4];
    if (threadIdx.x < 16) s_mvp[threadIdx.x] = mat4_mul_elem(p.model, p.light_space, threadIdx.x >> 2, threadIdx.x & 3);  // depth.vert:13
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool valid = t < p.n_tris;
    float clip[3][4] = {};
    if (valid) {
        const unsigned vmax = (unsigned)p.n_verts - 1u;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float *v = p.positions + 3 * (size_t)min(p.indices[3 * t + k], vmax);
            const float v4[4] = {v[0], v[1], v[2], 1.0f};
            mat4_vec4(s_mvp, v4, clip[k]);
        }
    }
    shadow_raster<false>(pools[threadIdx.x >> 6], valid, clip, clip, nullptr, 0.0f, p.res, false, 1.0f, p.out);
}

__global__ __launch_bounds__(256) void fill_depth_kernel(float4 *out, size_t n4, float value)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = make_float4(value, value, value, value);
}

static int32_t check_draws(const char *who, const VkrMeshDraw *draws, int32_t n_draws)
{
    VKR_REQUIRE(n_draws >= 0 && (n_draws == 0 || draws), "%s: draws missing", who);
    for (int d = 0; d < n_draws; d++)
        VKR_REQUIRE(draws[d].positions && draws[d].indices && draws[d].n_verts > 0 && draws[d].n_tris >= 0, "%s: draw %d has no mesh", who, d);
    return VKR_OK;
}

static void fill_params(ShadowParams &p, const VkrMeshDraw &d)
{
    p.positions = d.positions; p.indices = d.indices; p.n_verts = d.n_verts; p.n_tris = d.n_tris;
    for (int i = 0; i < 16; i++) p.model[i] = d.model[i];
}

}  // namespace vkr

extern "C" int32_t vkr_point_shadow_maps(const VkrPointShadowArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_point_shadow_maps: args is null");
    int32_t rc = check_draws("vkr_point_shadow_maps", a->draws, a->n_draws);
    if (rc != VKR_OK) return rc;
    VKR_REQUIRE(a->n_lights >= 0 && a->n_lights <= 65535 && (a->n_lights == 0 || a->lights), "vkr_point_shadow_maps: lights missing or too many");
    VKR_REQUIRE(a->res > 0 && a->res <= 8192 && a->res % 2 == 0, "vkr_point_shadow_maps: face size %d (even, at most 8192)", a->res);
    VKR_REQUIRE(a->far_plane > 1.0f, "vkr_point_shadow_maps: far plane %g must lie beyond the near plane at 1", (double)a->far_plane);
    VKR_REQUIRE(a->layout == VKR_CUBE_LAYOUT_LINEAR || ((a->layout == VKR_CUBE_LAYOUT_TILED_8X4 || a->layout == VKR_CUBE_LAYOUT_TILED_8X4_X25) && a->res % 8 == 0), "vkr_point_shadow_maps: layout %d not usable with face size %d", a->layout, a->res);
    VKR_REQUIRE(a->out != nullptr, "vkr_point_shadow_maps: output is null");
    if (a->n_lights == 0) return VKR_OK;
    const size_t n4 = (size_t)a->n_lights * 6 * a->res * a->res / 4;
    const float scale = a->layout == VKR_CUBE_LAYOUT_TILED_8X4_X25 ? 25.0f : 1.0f;
    hipLaunchKernelGGL(fill_depth_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), (float4 *)a->out, n4, scale);  // clear value 1.0
    ShadowParams p;
    p.lights = (const float *)a->lights;
    p.res = a->res; p.far_plane = a->far_plane; p.tiled = a->layout != VKR_CUBE_LAYOUT_LINEAR; p.store_scale = scale; p.out = a->out;
    for (int i = 0; i < 16; i++) p.light_space[i] = 0.0f;
    for (int d = 0; d < a->n_draws; d++) {
        if (a->draws[d].n_tris == 0) continue;
        fill_params(p, a->draws[d]);
        hipLaunchKernelGGL(point_shadow_kernel, dim3((unsigned)((p.n_tris + VKR_PS_BLOCK - 1) / VKR_PS_BLOCK), (unsigned)a->n_lights), dim3(VKR_PS_BLOCK), 0, as_stream(stream), p);
    }
    return launch_status("vkr_point_shadow_maps");
}

extern "C" int32_t vkr_dir_shadow_map(const VkrDirShadowArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_dir_shadow_map: args is null");
    int32_t rc = check_draws("vkr_dir_shadow_map", a->draws, a->n_draws);
    if (rc != VKR_OK) return rc;
    VKR_REQUIRE(a->res > 0 && a->res <= 16384 && a->res % 2 == 0, "vkr_dir_shadow_map: map size %d (even, at most 16384)", a->res);
    VKR_REQUIRE(a->out != nullptr, "vkr_dir_shadow_map: output is null");
    const size_t n4 = (size_t)a->res * a->res / 4;
    hipLaunchKernelGGL(fill_depth_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), (float4 *)a->out, n4, 1.0f);
    ShadowParams p;
    p.lights = nullptr;
    p.res = a->res; p.far_plane = 0.0f; p.tiled = 0; p.store_scale = 1.0f; p.out = a->out;
    for (int i = 0; i < 16; i++) p.light_space[i] = a->light_space[i];
    for (int d = 0; d < a->n_draws; d++) {
        if (a->draws[d].n_tris == 0) continue;
        fill_params(p, a->draws[d]);
        hipLaunchKernelGGL(dir_shadow_kernel, dim3((unsigned)((p.n_tris + 255) / 256)), dim3(256), 0, as_stream(stream), p);
    }
    return launch_status("vkr_dir_shadow_map");
}
