// Geometry pass on gfx950: gpass.vert / gpass.frag over the frame's draw calls
// (VulkanRenderer.cs:378-404), BACK faces culled, CCW front, depth test LESS, four RGBA16F
// targets cleared to (0,0,0,0), (0,0,0,1), (0,0,0,1), (0,0,0,1) (VulkanRenderer.cs:1155-1161,
// GraphicsPipeline.cs:71-73).
//
// Two kernels instead of a depth-tested raster pipeline:
//  1. visibility -- one lane per triangle walks the pixels of its bounding box and takes
//     atomicMin on a 64-bit key per pixel, depth bits (z/w as binary32) above the triangle's
//     number in submission order: the minimum is the LESS winner, an earlier triangle winning a
//     tie, whatever order lanes run in;
//  2. resolve -- one lane per pixel re-runs the vertex shader for the three corners of the
//     winner, interpolates its outputs perspective-correctly, samples the material and writes the
//     four targets once.  Hidden fragments are never shaded and no target is read back.
// Material textures are single-level RGBA8 images read through textureSampler (LINEAR, REPEAT;
// the sampler's anisotropic filtering is implementation-defined and not modelled); albedo and
// the ao/roughness/metalness image are _SRGB, the normal map _UNORM (Texture.cs:21).
#include <math.h>
#include <mutex>
#include "vkr_device.h"
#include "vkr_host.h"
#include "vkr_raster.h"

namespace vkr {

__device__ float g_gp_srgb[256];  // sRGB 8-bit -> linear, from the host C library like the oracle's
static std::mutex g_gp_mutex;
static bool g_gp_ready[64];

static int32_t ensure_srgb_table()
{
    int dev = 0;
    VKR_HIP_CHECK(hipGetDevice(&dev));
    VKR_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_gp_mutex);
    if (g_gp_ready[dev]) return VKR_OK;
    static float lut[256];
    for (int i = 0; i < 256; i++) {
        float c = (float)i * (1.0f / 255.0f);
        lut[i] = c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f);
    }
    VKR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gp_srgb), lut, sizeof(lut)));
    g_gp_ready[dev] = true;
    return VKR_OK;
}

struct GpassParams {
    const VkrGpassDraw *draws;  // device copy
    int n_draws;
    int draw;                   // visibility kernel: the draw of this launch
    unsigned tri_base;          // number of its first triangle in submission order
    unsigned scatter_stride;    // lane s of the launch takes triangle s * stride mod n (stride coprime to n)
    float vp[16];               // (proj * view), row-vector convention
    int width, rows, frame_rows, row0;
    unsigned long long *vis;
    uint2 *albedo, *normal, *arm, *position;
    struct GpTriRecord *tri_records;  // one per triangle in submission order, or null: the resolve step shades the corners per pixel
};

struct GpVertex { float pos[4], uv[2], tbn[9], clip[4]; };

// What the resolve step needs of a triangle: the edge functions of the flipped (front-facing) triangle and the
// vertex shader's outputs at its three corners.
struct alignas(16) GpTriRecord { double a[3], b[3], c[3]; float pos[3][4], uv[3][2], tbn[3][9], pad; };
static_assert(sizeof(GpTriRecord) == 256, "sixteen 16-byte loads per record");

// gpass.vert:24-35
VKR_DEV void gpass_vertex(const VkrGpassDraw &d, const float *vp, unsigned i, GpVertex &o, bool attributes)
{
    i = min(i, (unsigned)d.n_verts - 1u);  // an index past the arrays would fault
    const float p4[4] = {d.positions[3 * i], d.positions[3 * i + 1], d.positions[3 * i + 2], 1.0f};
    mat4_vec4(d.model, p4, o.pos);
    mat4_vec4(vp, o.pos, o.clip);
    if (!attributes) return;
    o.uv[0] = d.texcoords[2 * i]; o.uv[1] = d.texcoords[2 * i + 1];
    const float t4[4] = {d.tangents[3 * i], d.tangents[3 * i + 1], d.tangents[3 * i + 2], 0.0f};
    const float n4[4] = {d.normals[3 * i], d.normals[3 * i + 1], d.normals[3 * i + 2], 0.0f};
    float tw[4], nw[4];
    mat4_vec4(d.model, t4, tw);
    mat4_vec4(d.model, n4, nw);
    f3 T = normalize3(mk3(tw[0], tw[1], tw[2]));
    const f3 N = normalize3(mk3(nw[0], nw[1], nw[2]));
    const float tn = dot3(T, N);
    T = normalize3(mk3(T.x - tn * N.x, T.y - tn * N.y, T.z - tn * N.z));
    const f3 B = cross3(N, T);
    o.tbn[0] = T.x; o.tbn[1] = T.y; o.tbn[2] = T.z;
    o.tbn[3] = B.x; o.tbn[4] = B.y; o.tbn[5] = B.z;
    o.tbn[6] = N.x; o.tbn[7] = N.y; o.tbn[8] = N.z;
}

// The records of one draw's triangles: gpass.vert at the three corners and the set-up the resolve step would
// otherwise redo for every pixel (the same functions on the same inputs, so the same bits).
__global__ __launch_bounds__(256) void gpass_triangle_kernel(const GpassParams p)
{
    const VkrGpassDraw &d = p.draws[p.draw];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.n_tris) return;
    GpVertex v[3];
    float clip[3][4];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        gpass_vertex(d, p.vp, d.indices[3 * t + k], v[k], true);
#pragma unroll
        for (int j = 0; j < 4; j++) clip[k][j] = v[k].clip[j];
    }
    STri tr;
    raster_setup(clip, tr);
    raster_flip(tr);
    GpTriRecord &r = p.tri_records[p.tri_base + (unsigned)t];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.a[k] = tr.a[k]; r.b[k] = tr.b[k]; r.c[k] = tr.c[k];
#pragma unroll
        for (int j = 0; j < 4; j++) r.pos[k][j] = v[k].pos[j];
        r.uv[k][0] = v[k].uv[0]; r.uv[k][1] = v[k].uv[1];
#pragma unroll
        for (int j = 0; j < 9; j++) r.tbn[k][j] = v[k].tbn[j];
    }
    r.pad = 0.0f;
}

#ifndef VKR_VIS_SPREAD
#define VKR_VIS_SPREAD 4
#endif
static constexpr int VIS_SPREAD = VKR_VIS_SPREAD;

__global__ __launch_bounds__(256) void gpass_visibility_kernel(const GpassParams p)
{
    __shared__ RasterPool pools[4];
    const VkrGpassDraw &d = p.draws[p.draw];
    // Lane -> triangle: neighbours in a mesh are neighbours on the screen and of similar size, so with consecutive
    // triangles per wave the few waves that own the triangles next to the camera would run long after the rest
    // (there is about one wave per SIMD).  Triangle s * stride mod n for lane s spreads every wave over the mesh;
    // the stride is coprime to n, so every triangle is taken exactly once.
    // Only every VIS_SPREAD-th lane brings a triangle: with a lane each, a 66 000-triangle mesh is one wave per SIMD,
    // nothing hides the latency of a wave's binary64 chains and atomics, and the vector ALUs are 14 % busy; the samples
    // of all but the smallest boxes are walked by the whole wave anyway.
    const int g = blockIdx.x * 256 + threadIdx.x, s = g / VIS_SPREAD;
    bool have = g % VIS_SPREAD == 0 && s < d.n_tris;  // the other lanes stay: the wave shares large triangles
    const int t = have ? (int)(((unsigned long long)s * p.scatter_stride) % (unsigned)d.n_tris) : 0;
    float clip[3][4] = {};
    int bx0 = 0, by0 = 0, bx1 = -1, by1 = -1;
    if (have) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            GpVertex v;
            gpass_vertex(d, p.vp, d.indices[3 * t + k], v, false);
#pragma unroll
            for (int j = 0; j < 4; j++) clip[k][j] = v.clip[j];
        }
        STri tr;
        raster_setup(clip, tr);
        int ry0, ry1;
        // BACK faces (D > 0) and degenerate triangles are culled
        have = !clip_reject(clip) && tr.D < 0.0 && raster_bbox(clip, p.width, p.frame_rows, bx0, ry0, bx1, ry1);
        if (have) {
            by0 = max(ry0 - p.row0, 0); by1 = min(ry1 - p.row0, p.rows - 1);
            have = by0 <= by1;
        }
    }
    const double sx = 2.0 / (double)p.width, sy = 2.0 / (double)p.frame_rows;
    raster_boxes<true, true>(pools[threadIdx.x >> 6], have, clip, clip, p.tri_base + (unsigned)t, bx0, by0, bx1, by1, [&](const STri &tr, const float (&)[3][4], unsigned id, int x, int y) {
        const double xn = __builtin_fma((double)x + 0.5, sx, -1.0), yn = __builtin_fma((double)(p.row0 + y) + 0.5, sy, -1.0);
        const double e0 = s_edge(tr, 0, xn, yn), e1 = s_edge(tr, 1, xn, yn), e2 = s_edge(tr, 2, xn, yn);
        if (!(e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0 && (e0 + e1) + e2 > 0.0)) return;
        const double zs = __builtin_fma(e2, tr.z[2], __builtin_fma(e1, tr.z[1], e0 * tr.z[0]));
        if (!(zs >= 0.0 && zs <= tr.D)) return;
        const float depth = (float)(zs * tr.inv_D) + 0.0f;
        if (!(depth < 1.0f)) return;  // CompareOp.Less against the depth buffer's clear value 1.0
        const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned int, depth) << 32) | (unsigned long long)id;
        atomicMin(p.vis + (size_t)y * p.width + x, key);
    });
}

// textureSampler on a single-level RGBA8 image: LINEAR, REPEAT; srgb: r, g, b are decoded before filtering through
// srgb_lut, the workgroup's copy of g_gp_srgb in LDS
VKR_DEV void tex2d_rgba8(const VkrTexture &t, bool srgb, float s, float tt, const float *srgb_lut, float (&o)[4])
{
    const int w = t.width, h = t.height;
    const float x = s * (float)w - 0.5f, y = tt * (float)h - 0.5f;
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    const float al = x - fx, be = y - fy;
    const int i0 = wrap_repeat((int)fx, w), i1 = wrap_repeat((int)fx + 1, w);
    const int j0 = wrap_repeat((int)fy, h), j1 = wrap_repeat((int)fy + 1, h);
    const float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
    const uint32_t *img = (const uint32_t *)t.rgba8;
    const uint32_t p00 = img[(size_t)j0 * w + i0], p10 = img[(size_t)j0 * w + i1], p01 = img[(size_t)j1 * w + i0], p11 = img[(size_t)j1 * w + i1];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t b00 = (p00 >> (8 * c)) & 0xFF, b10 = (p10 >> (8 * c)) & 0xFF, b01 = (p01 >> (8 * c)) & 0xFF, b11 = (p11 >> (8 * c)) & 0xFF;
        float a, b, cc, d;
        if (srgb && c < 3) { a = srgb_lut[b00]; b = srgb_lut[b10]; cc = srgb_lut[b01]; d = srgb_lut[b11]; }
        else { a = (float)b00 * (1.0f / 255.0f); b = (float)b10 * (1.0f / 255.0f); cc = (float)b01 * (1.0f / 255.0f); d = (float)b11 * (1.0f / 255.0f); }
        o[c] = fma_(w11, d, fma_(w01, cc, fma_(w10, b, w00 * a)));
    }
}

// RECORDS: the winner's corners and edge functions are read from p.tri_records instead of being recomputed per pixel.
template <bool RECORDS>
__global__ __launch_bounds__(256) void gpass_resolve_kernel(const GpassParams p)
{
    __shared__ float s_srgb[256];
    s_srgb[threadIdx.x] = g_gp_srgb[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.width || y >= p.rows) return;
    const size_t px = (size_t)y * p.width + x;
    const unsigned long long key = p.vis[px];
    if (key == ~0ull) {  // clear values
        p.albedo[px] = make_uint2(0u, 0u);
        const uint2 zero_one = pack4(0.0f, 0.0f, 0.0f, 1.0f);
        p.normal[px] = zero_one; p.arm[px] = zero_one; p.position[px] = zero_one;
        return;
    }
    unsigned id = (unsigned)(key & 0xFFFFFFFFull), base = 0;
    int di = 0;
    while (di + 1 < p.n_draws && id - base >= (unsigned)p.draws[di].n_tris) { base += (unsigned)p.draws[di].n_tris; di++; }
    const VkrGpassDraw &d = p.draws[di];
    const unsigned t = id - base;
    const double sx = 2.0 / (double)p.width, sy = 2.0 / (double)p.frame_rows;
    const double xn = __builtin_fma((double)x + 0.5, sx, -1.0), yn = __builtin_fma((double)(p.row0 + y) + 0.5, sy, -1.0);
    GpVertex v[3];
    double e0, e1, e2;
    if (RECORDS) {
        const GpTriRecord &r = p.tri_records[id];
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[k].pos[j] = r.pos[k][j];
            v[k].uv[0] = r.uv[k][0]; v[k].uv[1] = r.uv[k][1];
#pragma unroll
            for (int j = 0; j < 9; j++) v[k].tbn[j] = r.tbn[k][j];
        }
        e0 = __builtin_fma(r.a[0], xn, __builtin_fma(r.b[0], yn, r.c[0]));
        e1 = __builtin_fma(r.a[1], xn, __builtin_fma(r.b[1], yn, r.c[1]));
        e2 = __builtin_fma(r.a[2], xn, __builtin_fma(r.b[2], yn, r.c[2]));
    } else {
        float clip[3][4];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            gpass_vertex(d, p.vp, d.indices[3 * t + k], v[k], true);
#pragma unroll
            for (int j = 0; j < 4; j++) clip[k][j] = v[k].clip[j];
        }
        STri tr;
        raster_setup(clip, tr);
        raster_flip(tr);
        e0 = s_edge(tr, 0, xn, yn); e1 = s_edge(tr, 1, xn, yn); e2 = s_edge(tr, 2, xn, yn);
    }
    // attributes are interpolated in binary32, as in the shadow pass: the binary64 edge values rounded to binary32 are the
    // perspective-correct weights
    const float f0 = (float)e0, f1 = (float)e1, f2 = (float)e2;
    const float inv_s = rcp_rn_checked((f0 + f1) + f2);  // one division per pixel
    auto lerp = [&](float a0, float a1, float a2) { return fma_(f2, a2, fma_(f1, a1, f0 * a0)) * inv_s; };
    const float u = lerp(v[0].uv[0], v[1].uv[0], v[2].uv[0]), vv = lerp(v[0].uv[1], v[1].uv[1], v[2].uv[1]);
    float pos[4], tbn[9];
#pragma unroll
    for (int k = 0; k < 4; k++) pos[k] = lerp(v[0].pos[k], v[1].pos[k], v[2].pos[k]);
#pragma unroll
    for (int k = 0; k < 9; k++) tbn[k] = lerp(v[0].tbn[k], v[1].tbn[k], v[2].tbn[k]);
    // gpass.frag:18-28
    float alb[4], nm[4], arm[4];
    tex2d_rgba8(d.albedo, true, u, vv, s_srgb, alb);
    tex2d_rgba8(d.normal, false, u, vv, s_srgb, nm);
    tex2d_rgba8(d.arm, true, u, vv, s_srgb, arm);
    const f3 n = normalize3(mk3(fma_(nm[0], 2.0f, -1.0f), fma_(nm[1], 2.0f, -1.0f), fma_(nm[2], 2.0f, -1.0f)));
    const float wn0 = fma_(tbn[6], n.z, fma_(tbn[3], n.y, tbn[0] * n.x));
    const float wn1 = fma_(tbn[7], n.z, fma_(tbn[4], n.y, tbn[1] * n.x));
    const float wn2 = fma_(tbn[8], n.z, fma_(tbn[5], n.y, tbn[2] * n.x));
    p.albedo[px] = pack4(alb[0], alb[1], alb[2], alb[3]);
    p.normal[px] = pack4(wn0, wn1, wn2, 1.0f);
    p.arm[px] = pack4(arm[0], arm[1], arm[2], arm[3]);
    p.position[px] = pack4(pos[0], pos[1], pos[2], pos[3]);
}

// Skybox draw of the composition pass (VulkanRenderer.cs:436-443; cubemap.vert, skybox.frag): the cube lookup
// depends only on the direction of the interpolated cube position, i.e. on the pixel's view ray
// (xn / proj[0][0], yn / proj[1][1], -1) turned into world space by the transposed view rotation.
struct SkyboxParams {
    const void *env; int env_n; int tiled;
    float view[16]; float ipx, ipy;
    int width, rows, frame_rows, row0;
    uint2 *out;
};

template <bool TILED>
__global__ __launch_bounds__(256) void skybox_kernel(const SkyboxParams p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.width || y >= p.rows) return;
    const float sx = 2.0f / (float)p.width, sy = 2.0f / (float)p.frame_rows;
    const float xn = fma_((float)x + 0.5f, sx, -1.0f), yn = fma_((float)(p.row0 + y) + 0.5f, sy, -1.0f);
    const float vx = xn * p.ipx, vy = yn * p.ipy, vz = -1.0f;
    const float *V = p.view;
    const f3 d = mk3(fma_(vz, V[2], fma_(vy, V[1], vx * V[0])), fma_(vz, V[6], fma_(vy, V[5], vx * V[4])), fma_(vz, V[10], fma_(vy, V[9], vx * V[8])));
    const f3 c = texcube_rgb_linear<TILED>(p.env, p.env_n, d);
    p.out[(size_t)y * p.width + x] = pack4(c.x, c.y, c.z, 1.0f);
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *out, size_t n, unsigned long long value)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = value;
}

int32_t fill_u64(void *dst, size_t n, unsigned long long value, void *stream)
{
    if (n == 0) return VKR_OK;
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), (unsigned long long *)dst, n, value);
    return launch_status("fill");
}

}  // namespace vkr

extern "C" int64_t vkr_gbuffer_workspace_bytes(int32_t width, int32_t rows, int32_t n_draws)
{
    if (width <= 0 || rows <= 0 || n_draws < 0) return -1;
    return (int64_t)width * rows * 8 + (int64_t)n_draws * (int64_t)sizeof(VkrGpassDraw) + 256;
}

extern "C" int64_t vkr_gbuffer_workspace_bytes_with_triangles(int32_t width, int32_t rows, int32_t n_draws, int64_t n_triangles)
{
    const int64_t base = vkr_gbuffer_workspace_bytes(width, rows, n_draws);
    if (base < 0 || n_triangles < 0 || n_triangles >= (1ll << 32) - 1) return -1;
    return ((base + 255) & ~(int64_t)255) + n_triangles * (int64_t)sizeof(vkr::GpTriRecord);
}

namespace vkr {

// Checks everything about a geometry-pass call except where its scratch memory comes from.
static int32_t check_gbuffer(const char *who, const VkrGpassArgs *a)
{
    VKR_REQUIRE(a != nullptr, "%s: args is null", who);
    VKR_REQUIRE(a->width > 0 && a->rows > 0, "%s: bad size %dx%d", who, a->width, a->rows);
    VKR_REQUIRE(a->frame_rows >= a->rows && a->row0 >= 0 && a->row0 + a->rows <= a->frame_rows, "%s: rows [%d,%d) outside a frame of %d rows", who, a->row0, a->row0 + a->rows, a->frame_rows);
    VKR_REQUIRE(a->n_draws >= 0 && (a->n_draws == 0 || a->draws), "%s: draws missing", who);
    VKR_REQUIRE(a->albedo && a->normal && a->arm && a->position, "%s: an output image is null", who);
    uint64_t total_tris = 0;
    for (int d = 0; d < a->n_draws; d++) {
        const VkrGpassDraw &g = a->draws[d];
        VKR_REQUIRE(g.positions && g.texcoords && g.normals && g.tangents && g.indices && g.n_verts > 0 && g.n_tris >= 0, "%s: draw %d has no mesh", who, d);
        const VkrTexture *tex[3] = {&g.albedo, &g.normal, &g.arm};
        for (const VkrTexture *t : tex) VKR_REQUIRE(t->rgba8 && t->width > 0 && t->height > 0, "%s: draw %d lacks a material image", who, d);
        total_tris += (uint64_t)g.n_tris;
    }
    VKR_REQUIRE(total_tris < (1ull << 32) - 1, "%s: too many triangles", who);
    return VKR_OK;
}

// The launches of the geometry pass.  `vis` (one 64-bit key per pixel) and `dev_draws` (room for the draw
// records) are device scratch memory; every key is reset here, so the scratch may hold anything on entry.
static int32_t launch_gbuffer(const char *who, const VkrGpassArgs *a, unsigned long long *vis, VkrGpassDraw *dev_draws, GpTriRecord *tri_records, void *stream)
{
    GpassParams p;
    p.tri_records = tri_records;
    const size_t npx = (size_t)a->width * a->rows;
    p.vis = vis;
    p.draws = dev_draws; p.n_draws = a->n_draws;
    if (a->n_draws) VKR_HIP_CHECK(hipMemcpyAsync(dev_draws, a->draws, (size_t)a->n_draws * sizeof(VkrGpassDraw), hipMemcpyHostToDevice, as_stream(stream)));
    // (proj * view), as gpass.vert:26 associates it
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            const float *v = a->scene.camera_view, *q = a->scene.camera_projection;
            p.vp[4 * i + j] = fmaf(v[4 * i + 3], q[12 + j], fmaf(v[4 * i + 2], q[8 + j], fmaf(v[4 * i + 1], q[4 + j], v[4 * i] * q[j])));
        }
    p.width = a->width; p.rows = a->rows; p.frame_rows = a->frame_rows; p.row0 = a->row0;
    p.albedo = (uint2 *)a->albedo; p.normal = (uint2 *)a->normal; p.arm = (uint2 *)a->arm; p.position = (uint2 *)a->position;
    p.draw = 0; p.tri_base = 0; p.scatter_stride = 1;
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, as_stream(stream), p.vis, npx, ~0ull);
    unsigned base = 0;
    for (int d = 0; d < a->n_draws; d++) {
        if (a->draws[d].n_tris > 0) {
            p.draw = d; p.tri_base = base;
            p.scatter_stride = scatter_stride_for((unsigned)a->draws[d].n_tris);
            hipLaunchKernelGGL(gpass_visibility_kernel, dim3((unsigned)(((size_t)a->draws[d].n_tris * VIS_SPREAD + 255) / 256)), dim3(256), 0, as_stream(stream), p);
            if (tri_records) hipLaunchKernelGGL(gpass_triangle_kernel, dim3((unsigned)((a->draws[d].n_tris + 255) / 256)), dim3(256), 0, as_stream(stream), p);
        }
        base += (unsigned)a->draws[d].n_tris;
    }
    const dim3 grid((unsigned)((a->width + 63) / 64), (unsigned)((a->rows + 3) / 4));
    if (tri_records) hipLaunchKernelGGL(gpass_resolve_kernel<true>, grid, dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(gpass_resolve_kernel<false>, grid, dim3(256), 0, as_stream(stream), p);
    return launch_status(who);
}

// Scratch memory of vkr_gbuffer_static: the visibility keys followed by the draw records, resident in every
// device's copy of the code object from the moment the library is loaded (16 MiB + 16 KiB per GPU).
static constexpr size_t GPS_KEY_BYTES = (size_t)VKR_GBUFFER_STATIC_MAX_PIXELS * 8;
static constexpr size_t GPS_DRAW_BYTES = ((size_t)VKR_GBUFFER_STATIC_MAX_DRAWS * sizeof(VkrGpassDraw) + 255) & ~(size_t)255;
static __device__ __align__(256) unsigned char scratch[GPS_KEY_BYTES + GPS_DRAW_BYTES];
static void *g_scratch_ptr[64];  // per device ordinal: the device address of that GPU's copy of `scratch`

// The device address of the current device's `scratch`, looked up once per device.  (In host code the
// array's name is only a handle for the symbol; kernels and copies must be given this address.)
static int32_t static_scratch(void **out)
{
    int dev = 0;
    VKR_HIP_CHECK(hipGetDevice(&dev));
    VKR_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_gp_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        VKR_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        VKR_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return VKR_OK;
}

}  // namespace vkr

extern "C" int32_t vkr_gbuffer(const VkrGpassArgs *a, void *stream)
{
    using namespace vkr;
    int32_t rc = check_gbuffer("vkr_gbuffer", a);
    if (rc != VKR_OK) return rc;
    const int64_t need = vkr_gbuffer_workspace_bytes(a->width, a->rows, a->n_draws);
    VKR_REQUIRE(a->workspace && a->workspace_bytes >= need, "vkr_gbuffer: workspace of %lld bytes needed (vkr_gbuffer_workspace_bytes)", (long long)need);
    if ((rc = ensure_srgb_table()) != VKR_OK) return rc;
    const size_t npx = (size_t)a->width * a->rows;
    VkrGpassDraw *dev_draws = (VkrGpassDraw *)((char *)a->workspace + ((npx * 8 + 255) & ~(size_t)255));
    // a workspace of vkr_gbuffer_workspace_bytes_with_triangles() also has room for a record per triangle
    int64_t total_tris = 0;
    for (int d = 0; d < a->n_draws; d++) total_tris += a->draws[d].n_tris;
    const int64_t with_records = vkr_gbuffer_workspace_bytes_with_triangles(a->width, a->rows, a->n_draws, total_tris);
    GpTriRecord *tri_records = nullptr;
    if (total_tris > 0 && with_records > 0 && a->workspace_bytes >= with_records)
        tri_records = (GpTriRecord *)((char *)a->workspace + (((size_t)need + 255) & ~(size_t)255));
    return launch_gbuffer("vkr_gbuffer", a, (unsigned long long *)a->workspace, dev_draws, tri_records, stream);
}

extern "C" int32_t vkr_gbuffer_static(const VkrGpassArgs *a, void *stream)
{
    using namespace vkr;
    int32_t rc = check_gbuffer("vkr_gbuffer_static", a);
    if (rc != VKR_OK) return rc;
    VKR_REQUIRE((int64_t)a->width * a->rows <= (int64_t)VKR_GBUFFER_STATIC_MAX_PIXELS,
                "vkr_gbuffer_static: %dx%d = %lld pixels exceed the fixed scratch of %d pixels; use vkr_gbuffer with a workspace",
                a->width, a->rows, (long long)a->width * a->rows, VKR_GBUFFER_STATIC_MAX_PIXELS);
    VKR_REQUIRE(a->n_draws <= VKR_GBUFFER_STATIC_MAX_DRAWS, "vkr_gbuffer_static: %d draws exceed the fixed scratch of %d draws; use vkr_gbuffer with a workspace",
                a->n_draws, VKR_GBUFFER_STATIC_MAX_DRAWS);
    if ((rc = ensure_srgb_table()) != VKR_OK) return rc;
    void *base = nullptr;
    if ((rc = static_scratch(&base)) != VKR_OK) return rc;
    return launch_gbuffer("vkr_gbuffer_static", a, (unsigned long long *)base, (VkrGpassDraw *)((char *)base + GPS_KEY_BYTES), nullptr, stream);
}

extern "C" int32_t vkr_skybox(const void *env, int32_t env_n, int32_t env_layout, const VkrSceneInfo *scene,
                              int32_t width, int32_t rows, int32_t frame_rows, int32_t row0, void *out_rgba16f, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(env && scene && out_rgba16f && env_n > 0, "vkr_skybox: an image or the scene is null");
    VKR_REQUIRE(width > 0 && rows > 0 && frame_rows >= rows && row0 >= 0 && row0 + rows <= frame_rows, "vkr_skybox: bad size %dx%d, rows [%d,%d) of %d", width, rows, row0, row0 + rows, frame_rows);
    VKR_REQUIRE(env_layout == VKR_ENV_LAYOUT_LINEAR || (env_layout == VKR_ENV_LAYOUT_TILED_4X4 && env_n % 4 == 0), "vkr_skybox: layout %d not usable with face size %d", env_layout, env_n);
    VKR_REQUIRE(scene->camera_projection[0] != 0.0f && scene->camera_projection[5] != 0.0f, "vkr_skybox: the projection matrix has a zero scale");
    SkyboxParams p;
    p.env = env; p.env_n = env_n; p.tiled = env_layout == VKR_ENV_LAYOUT_TILED_4X4;
    for (int i = 0; i < 16; i++) p.view[i] = scene->camera_view[i];
    p.ipx = 1.0f / scene->camera_projection[0]; p.ipy = 1.0f / scene->camera_projection[5];
    p.width = width; p.rows = rows; p.frame_rows = frame_rows; p.row0 = row0;
    p.out = (uint2 *)out_rgba16f;
    const dim3 grid((unsigned)((width + 63) / 64), (unsigned)((rows + 3) / 4));
    if (p.tiled) hipLaunchKernelGGL(skybox_kernel<true>, grid, dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(skybox_kernel<false>, grid, dim3(256), 0, as_stream(stream), p);
    return launch_status("vkr_skybox");
}
