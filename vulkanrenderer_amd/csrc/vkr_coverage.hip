// Light-volume coverage on gfx950: which pixels does the light.frag draw of each light touch?
//
// The reference draws, per light, a sphere mesh scaled to the light's reach (Light.cs:10-26)
// through light.vert:26 with FRONT faces culled, CCW front, no depth test
// (GraphicsPipeline.cs:100-117); the fixed-function rasteriser decides the pixels.  Here a pixel
// belongs to the draw when its centre lies inside the projection of a back-facing triangle and
// the triangle's depth there passes the 0 <= z <= w clip.  Evaluated in homogeneous form, so a
// triangle that crosses the eye plane needs no clipping: with V0,V1,V2 the clip-space (x,y,w) of
// the corners and p = (xn, yn, 1) the pixel centre in normalised device coordinates,
//   E0 = p.(V1xV2)  E1 = p.(V2xV0)  E2 = p.(V0xV1)  D = V0.(V1xV2)
//   back face: D > 0;  inside, in front of the eye: E0,E1,E2 >= 0 and E0+E1+E2 > 0;
//   depth clip: 0 <= E0*z0 + E1*z1 + E2*z2 <= D.
// Everything after the binary32 vertex transform is binary64 with one fixed operation order, so
// the masks do not depend on how the frame is cut into blocks, strips or GPUs.
//
// Work layout: one workgroup per (light, 128x128-pixel block).  It transforms the mesh into LDS,
// sets up 256 triangles at a time (one per lane), keeps those whose edges do not rule the block
// out, and tests the block's pixels against the kept ones: a wave walks 64 of the block's 8x8
// tiles with lane = pixel, so a ballot is the tile's 64-bit mask.
#include "vkr_device.h"
#include "vkr_host.h"

namespace vkr {

static constexpr int COV_BLOCK = 128;      // pixels per side of a workgroup's block
static constexpr int COV_MAX_VERTS = 1024; // clip-space vertices held in LDS

struct CoverageParams {
    int width, rows, frame_rows, row0;
    float view[16], proj[16];
    const float *lights;  // VkrLightInfo array viewed as floats, 23 per light
    int n_lights;
    const float *verts; int n_verts;
    const unsigned *indices; int n_tris;
    unsigned long long *masks;
    unsigned *flags;  // per light: 1 = covers every pixel (coverage_classify_kernel); or null
    int tiles_x, tiles_y, blocks_x;
};

struct CovTri { double a[3], b[3], c[3], z[3], D; };

VKR_DEV double cov_edge(const CovTri &t, int i, double xn, double yn) { return __builtin_fma(t.a[i], xn, __builtin_fma(t.b[i], yn, t.c[i])); }

// light.vert:26 for every vertex of the volume mesh of light `li`, evaluated left to right: (proj * view) * model,
// row-vector convention.  Called by all 256 threads of a workgroup; the clip-space vertices are in LDS when it returns.
VKR_DEV void transform_volume(const CoverageParams &p, int li, float *s_vp, float *s_mvp, float4 *s_clip)
{
    const int tid = threadIdx.x;
    const float *model = p.lights + (size_t)li * 23;
    if (tid < 16) {
        const int i = tid >> 2, j = tid & 3;
        s_vp[tid] = fma_(p.view[4 * i + 3], p.proj[12 + j], fma_(p.view[4 * i + 2], p.proj[8 + j], fma_(p.view[4 * i + 1], p.proj[4 + j], p.view[4 * i] * p.proj[j])));
    }
    __syncthreads();
    if (tid < 16) {
        const int i = tid >> 2, j = tid & 3;
        s_mvp[tid] = fma_(model[4 * i + 3], s_vp[12 + j], fma_(model[4 * i + 2], s_vp[8 + j], fma_(model[4 * i + 1], s_vp[4 + j], model[4 * i] * s_vp[j])));
    }
    __syncthreads();
    for (int v = tid; v < p.n_verts; v += 256) {
        const float x = p.verts[3 * v], y = p.verts[3 * v + 1], z = p.verts[3 * v + 2];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = fma_(z, s_mvp[8 + j], fma_(y, s_mvp[4 + j], x * s_mvp[j])) + s_mvp[12 + j];
        s_clip[v] = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
}

// Edge functions and the back-face determinant of one triangle from its clip-space corners.
VKR_DEV void cov_setup(const float4 &c0, const float4 &c1, const float4 &c2, CovTri &tr)
{
    const double x0 = c0.x, y0 = c0.y, w0 = c0.w, x1 = c1.x, y1 = c1.y, w1 = c1.w, x2 = c2.x, y2 = c2.y, w2 = c2.w;
    tr.a[0] = y1 * w2 - w1 * y2; tr.b[0] = w1 * x2 - x1 * w2; tr.c[0] = x1 * y2 - y1 * x2;
    tr.a[1] = y2 * w0 - w2 * y0; tr.b[1] = w2 * x0 - x2 * w0; tr.c[1] = x2 * y0 - y2 * x0;
    tr.a[2] = y0 * w1 - w0 * y1; tr.b[2] = w0 * x1 - x0 * w1; tr.c[2] = x0 * y1 - y0 * x1;
    tr.z[0] = c0.z; tr.z[1] = c1.z; tr.z[2] = c2.z;
    tr.D = __builtin_fma(w0, tr.c[0], __builtin_fma(y0, tr.b[0], x0 * tr.a[0]));
}

// Does light blockIdx.x cover every pixel?  Sufficient conditions, decided on the very clip-space vertices and
// determinants the rasterising kernel works with:
//  (1) every triangle with an area is seen from behind (D > 0): the camera lies inside the closed mesh, on the inner
//      side of every face, so every view ray leaves the mesh through exactly one triangle, and that triangle is one
//      the draw keeps (FRONT faces culled);
//  (2) every triangle is either in front of the near plane with all three corners (z >= 0, w > 0), or cannot be hit by
//      the ray of any pixel (all corners outside one side plane of the view volume, or all behind the eye): the
//      triangle a ray leaves through passes the z >= 0 half of the depth clip (a sum of non-negative products);
//  (3) every vertex is on the near side of the far plane by a relative margin of 1e-6, nine orders of magnitude above the
//      rounding of the binary64 depth test: the z <= w half passes.
// A triangle with two equal corners has no area: it touches no pixel and bounds no hole.  Anything else (a camera
// outside or on the mesh, a volume that reaches the far plane, a triangle that straddles the near plane inside the
// view volume) leaves the flag at 0 and the light is rasterised.
__global__ __launch_bounds__(256) void coverage_classify_kernel(const CoverageParams p)
{
    __shared__ float s_vp[16], s_mvp[16];
    __shared__ float4 s_clip[COV_MAX_VERTS];
    const int tid = threadIdx.x, li = blockIdx.x;
    transform_volume(p, li, s_vp, s_mvp, s_clip);
    bool bad = false;
    for (int v = tid; v < p.n_verts; v += 256) {
        const float4 c = s_clip[v];
        bad = bad || !(c.w - c.z > 1e-6f * fmaxf(fabsf(c.w), fabsf(c.z)));  // also catches NaN
    }
    const unsigned vmax = (unsigned)p.n_verts - 1u;
    for (int t = tid; t < p.n_tris; t += 256) {
        const float4 c0 = s_clip[min(p.indices[3 * t], vmax)], c1 = s_clip[min(p.indices[3 * t + 1], vmax)], c2 = s_clip[min(p.indices[3 * t + 2], vmax)];
        auto same = [](const float4 &a, const float4 &b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; };
        if (same(c0, c1) || same(c1, c2) || same(c2, c0)) continue;
        CovTri tr;
        cov_setup(c0, c1, c2, tr);
        const bool in_front = c0.z >= 0.0f && c1.z >= 0.0f && c2.z >= 0.0f && c0.w > 0.0f && c1.w > 0.0f && c2.w > 0.0f;
        const bool unseen = (c0.x > c0.w && c1.x > c1.w && c2.x > c2.w) || (c0.x < -c0.w && c1.x < -c1.w && c2.x < -c2.w) ||
                            (c0.y > c0.w && c1.y > c1.w && c2.y > c2.w) || (c0.y < -c0.w && c1.y < -c1.w && c2.y < -c2.w) ||
                            (c0.w < 0.0f && c1.w < 0.0f && c2.w < 0.0f);
        bad = bad || !(tr.D > 0.0) || !(in_front || unseen);
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0) {
        p.flags[li] = any_bad ? 0u : 1u;
        if (any_bad) atomicAdd(&p.flags[p.n_lights], 1u);
    }
}

__global__ __launch_bounds__(256) void coverage_kernel(const CoverageParams p)
{
    __shared__ float s_vp[16], s_mvp[16];
    __shared__ float4 s_clip[COV_MAX_VERTS];
    __shared__ CovTri s_tri[256];
    __shared__ int s_count;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = blockIdx.y;
    const int bx0 = (blockIdx.x % p.blocks_x) * COV_BLOCK, by0 = (blockIdx.x / p.blocks_x) * COV_BLOCK;
    const int bx1 = min(bx0 + COV_BLOCK, p.width), by1 = min(by0 + COV_BLOCK, p.rows);

    if (p.flags && p.flags[li]) {
        // The light covers every pixel (coverage_classify_kernel): every tile of the block gets its pixels inside
        // the image, which is what the rasterising path below arrives at triangle by triangle.
        const int tx = (bx0 >> 3) + (tid & 15), ty = (by0 >> 3) + (tid >> 4);
        if (tx < p.tiles_x && ty < p.tiles_y) {
            const int cols = min(8, p.width - 8 * tx), rws = min(8, p.rows - 8 * ty);
            const unsigned long long row_bits = (1ull << cols) - 1ull;  // cols <= 8
            unsigned long long m = 0;
            for (int r = 0; r < rws; r++) m |= row_bits << (8 * r);
            p.masks[((size_t)li * p.tiles_y + ty) * p.tiles_x + tx] = m;
        }
        return;
    }
    transform_volume(p, li, s_vp, s_mvp, s_clip);

    const double sx = 2.0 / (double)p.width, sy = 2.0 / (double)p.frame_rows;
    // centres of the block's corner pixels
    const double xa = __builtin_fma((double)bx0 + 0.5, sx, -1.0), xb = __builtin_fma((double)(bx1 - 1) + 0.5, sx, -1.0);
    const double ya = __builtin_fma((double)(p.row0 + by0) + 0.5, sy, -1.0), yb = __builtin_fma((double)(p.row0 + by1 - 1) + 0.5, sy, -1.0);
    // this lane's pixel inside each of the wave's 64 tiles: tile k of wave w is tile w*64 + k of the block's 16x16
    const int lx = lane & 7, ly = lane >> 3;
    unsigned long long covered = 0;  // bit k: this lane's pixel of tile k is covered

    // the indices of the next round's triangle are fetched while this round is worked on: the kernel is bound by
    // the latency of its few dependent loads, not by arithmetic
    unsigned nidx[3] = {0u, 0u, 0u};
    if (tid < p.n_tris) { nidx[0] = p.indices[3 * tid]; nidx[1] = p.indices[3 * tid + 1]; nidx[2] = p.indices[3 * tid + 2]; }
    for (int t0 = 0; t0 < p.n_tris; t0 += 256) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        const int t = t0 + tid;
        const unsigned idx[3] = {nidx[0], nidx[1], nidx[2]};
        if (t + 256 < p.n_tris) { nidx[0] = p.indices[3 * (t + 256)]; nidx[1] = p.indices[3 * (t + 256) + 1]; nidx[2] = p.indices[3 * (t + 256) + 2]; }
        if (t < p.n_tris) {
            // an index past the vertex array would read outside LDS: clamp it
            const unsigned vmax = (unsigned)p.n_verts - 1u;
            const float4 c0 = s_clip[min(idx[0], vmax)], c1 = s_clip[min(idx[1], vmax)], c2 = s_clip[min(idx[2], vmax)];
            // Cheap binary32 rejections before the binary64 set-up (they only ever drop triangles the exact test
            // below would drop for every pixel of the block): the triangle lies outside one plane of the clip
            // volume, or -- all three corners in front of the eye -- its bounding box in normalised device
            // coordinates, widened by 1e-4 (a fifth of a 4K pixel; the divisions are good to 1e-6), misses the block.
            bool reject = (c0.x > c0.w && c1.x > c1.w && c2.x > c2.w) || (c0.x < -c0.w && c1.x < -c1.w && c2.x < -c2.w) ||
                          (c0.y > c0.w && c1.y > c1.w && c2.y > c2.w) || (c0.y < -c0.w && c1.y < -c1.w && c2.y < -c2.w) ||
                          (c0.z < 0.0f && c1.z < 0.0f && c2.z < 0.0f);
            if (!reject && c0.w > 1e-6f && c1.w > 1e-6f && c2.w > 1e-6f) {
                const float i0 = 1.0f / c0.w, i1 = 1.0f / c1.w, i2 = 1.0f / c2.w;
                const float ax = c0.x * i0, bx = c1.x * i1, cx = c2.x * i2, ay = c0.y * i0, by = c1.y * i1, cy = c2.y * i2;
                const float pad = 1e-4f + 1e-5f * fmaxf(fmaxf(fabsf(ax), fabsf(bx)), fmaxf(fmaxf(fabsf(cx), fabsf(ay)), fmaxf(fabsf(by), fabsf(cy))));
                reject = fminf(ax, fminf(bx, cx)) - pad > (float)xb || fmaxf(ax, fmaxf(bx, cx)) + pad < (float)xa ||
                         fminf(ay, fminf(by, cy)) - pad > (float)yb || fmaxf(ay, fmaxf(by, cy)) + pad < (float)ya;
            }
            if (reject) goto next_triangle;
            {
            CovTri tr;
            cov_setup(c0, c1, c2, tr);
            bool keep = tr.D > 0.0;  // back face; front faces and degenerate triangles are culled
#pragma unroll
            for (int i = 0; i < 3; i++)  // an edge with all four corners outside rules the block out
                keep = keep && !(cov_edge(tr, i, xa, ya) < 0.0 && cov_edge(tr, i, xb, ya) < 0.0 && cov_edge(tr, i, xa, yb) < 0.0 && cov_edge(tr, i, xb, yb) < 0.0);
            if (keep) s_tri[atomicAdd(&s_count, 1)] = tr;
            }
        next_triangle:;
        }
        __syncthreads();
        const int count = s_count;
        // tile `lane` of this wave (for the per-tile corner test below)
        const int ktile = wave * 64 + lane;
        const int kx0 = bx0 + (ktile & 15) * 8, ky0 = by0 + (ktile >> 4) * 8;
        const int kx1 = min(kx0 + 8, bx1), ky1 = min(ky0 + 8, by1);
        const bool ktile_in_image = kx0 < bx1 && ky0 < by1;
        const double cxa = __builtin_fma((double)kx0 + 0.5, sx, -1.0), cxb = __builtin_fma((double)(kx1 - 1) + 0.5, sx, -1.0);
        const double cya = __builtin_fma((double)(p.row0 + ky0) + 0.5, sy, -1.0), cyb = __builtin_fma((double)(p.row0 + ky1 - 1) + 0.5, sy, -1.0);
        for (int e = 0; e < count; e++) {
            const CovTri &tr = s_tri[e];
            // lane = tile: the corner test drops the tile, four covered corners fill it (the edge
            // functions and the depth are linear over the tile); the rest is tested per pixel
            bool drop = !ktile_in_image, fill = ktile_in_image;
            {
                bool in[4] = {true, true, true, true};
                double es[4][3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    es[0][i] = cov_edge(tr, i, cxa, cya); es[1][i] = cov_edge(tr, i, cxb, cya);
                    es[2][i] = cov_edge(tr, i, cxa, cyb); es[3][i] = cov_edge(tr, i, cxb, cyb);
                    drop = drop || (es[0][i] < 0.0 && es[1][i] < 0.0 && es[2][i] < 0.0 && es[3][i] < 0.0);
#pragma unroll
                    for (int c = 0; c < 4; c++) in[c] = in[c] && es[c][i] >= 0.0;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const double zs = __builtin_fma(es[c][2], tr.z[2], __builtin_fma(es[c][1], tr.z[1], es[c][0] * tr.z[0]));
                    fill = fill && in[c] && (es[c][0] + es[c][1]) + es[c][2] > 0.0 && zs >= 0.0 && zs <= tr.D;
                }
            }
            fill = fill && !drop;
            covered |= __ballot(fill);  // lane = pixel again: every pixel of a filled tile (pixels outside the image are masked at the end)
            unsigned long long partial = __ballot(!drop && !fill);
            while (partial) {  // wave-uniform walk over the tiles that need per-pixel tests
                const int k = __builtin_ctzll(partial);
                partial &= partial - 1;
                const int tile = wave * 64 + k;
                const int x = bx0 + (tile & 15) * 8 + lx, y = by0 + (tile >> 4) * 8 + ly;
                const bool todo = x < bx1 && y < by1 && !((covered >> k) & 1ull);
                if (!__any(todo)) continue;
                const double xn = __builtin_fma((double)x + 0.5, sx, -1.0), yn = __builtin_fma((double)(p.row0 + y) + 0.5, sy, -1.0);
                const double e0 = cov_edge(tr, 0, xn, yn), e1 = cov_edge(tr, 1, xn, yn), e2 = cov_edge(tr, 2, xn, yn);
                bool in = todo && e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0 && (e0 + e1) + e2 > 0.0;
                if (__any(in)) {
                    const double zs = __builtin_fma(e2, tr.z[2], __builtin_fma(e1, tr.z[1], e0 * tr.z[0]));
                    in = in && zs >= 0.0 && zs <= tr.D;
                    if (in) covered |= 1ull << k;
                }
            }
        }
        __syncthreads();  // the list is rebuilt by the next round
    }

    // bit k of `covered` in lane l -> bit l of tile k's mask: 64 ballots, lane k keeping the k-th, then one
    // store per lane instead of 64 single-lane stores per wave
    unsigned long long *masks = p.masks + (size_t)li * p.tiles_x * p.tiles_y;
    unsigned long long mine = 0;
    for (int k = 0; k < 64; k++) {
        const int tile = wave * 64 + k;
        const int x = bx0 + (tile & 15) * 8 + lx, y = by0 + (tile >> 4) * 8 + ly;
        const unsigned long long m = __ballot(((covered >> k) & 1ull) && x < bx1 && y < by1);
        if (lane == k) mine = m;
    }
    {
        const int tile = wave * 64 + lane;
        const int tx = (bx0 >> 3) + (tile & 15), ty = (by0 >> 3) + (tile >> 4);
        if (tx < p.tiles_x && ty < p.tiles_y) masks[(size_t)ty * p.tiles_x + tx] = mine;
    }
}

}  // namespace vkr

extern "C" int32_t vkr_light_coverage(const VkrCoverageArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_light_coverage: args is null");
    VKR_REQUIRE(a->width > 0 && a->rows > 0, "vkr_light_coverage: bad size %dx%d", a->width, a->rows);
    VKR_REQUIRE(a->frame_rows >= a->rows && a->row0 >= 0 && a->row0 + a->rows <= a->frame_rows, "vkr_light_coverage: rows [%d,%d) outside a frame of %d rows", a->row0, a->row0 + a->rows, a->frame_rows);
    VKR_REQUIRE(a->n_lights >= 0 && (a->n_lights == 0 || a->lights), "vkr_light_coverage: lights missing");
    VKR_REQUIRE(a->verts && a->indices && a->n_verts > 0 && a->n_tris > 0, "vkr_light_coverage: light-volume mesh missing");
    VKR_REQUIRE(a->n_verts <= COV_MAX_VERTS, "vkr_light_coverage: %d vertices; at most %d are supported", a->n_verts, COV_MAX_VERTS);
    VKR_REQUIRE(a->masks != nullptr, "vkr_light_coverage: output is null");
    if (a->n_lights == 0) return VKR_OK;
    VKR_REQUIRE(a->n_lights <= 65535, "vkr_light_coverage: too many lights for one call");

    CoverageParams p;
    p.width = a->width; p.rows = a->rows; p.frame_rows = a->frame_rows; p.row0 = a->row0;
    for (int i = 0; i < 16; i++) { p.view[i] = a->scene.camera_view[i]; p.proj[i] = a->scene.camera_projection[i]; }
    p.lights = (const float *)a->lights; p.n_lights = a->n_lights;
    p.flags = a->light_flags;
    p.verts = a->verts; p.n_verts = a->n_verts;
    p.indices = a->indices; p.n_tris = a->n_tris;
    p.masks = (unsigned long long *)a->masks;
    p.tiles_x = (a->width + 7) / 8; p.tiles_y = (a->rows + 7) / 8;
    p.blocks_x = (a->width + COV_BLOCK - 1) / COV_BLOCK;
    const int blocks_y = (a->rows + COV_BLOCK - 1) / COV_BLOCK;
    if (p.flags) {
        VKR_HIP_CHECK(hipMemsetAsync(p.flags + a->n_lights, 0, sizeof(unsigned), as_stream(stream)));
        hipLaunchKernelGGL(coverage_classify_kernel, dim3((unsigned)a->n_lights), dim3(256), 0, as_stream(stream), p);
    }
    hipLaunchKernelGGL(coverage_kernel, dim3((unsigned)(p.blocks_x * blocks_y), (unsigned)a->n_lights), dim3(256), 0, as_stream(stream), p);
    return launch_status("vkr_light_coverage");
}
