// Post chain on gfx950: bloom_blur.frag (two separable 9-tap passes, VulkanRenderer.cs:504-540)
// and postprocess.frag (ACES + gamma, VulkanRenderer.cs:544-557).  All three passes move 16-20
// bytes per pixel and do a few dozen flops on them, so they are sized for HBM bandwidth:
// every input texel is fetched from memory once per workgroup, 8-16 bytes per lane.
#include "vkr_device.h"
#include "vkr_fastmath.h"
#include "vkr_host.h"

namespace vkr {

// bloom_blur.frag:15
static constexpr float BW0 = 0.227027f, BW1 = 0.1945946f, BW2 = 0.1216216f, BW3 = 0.054054f, BW4 = 0.016216f;

struct BloomParams {
    const uint2 *src; uint2 *dst;
    int width, rows, src_rows, src_row_offset, wrap, clamp_lo, clamp_hi;
};

VKR_DEV int wrap_index(int i, int n, int wrap, int lo, int hi)
{
    if (wrap == VKR_WRAP_REPEAT) return wrap_repeat(i, n);
    return min(max(i, lo), hi);
}

// Accumulation order of bloom_blur.frag:17-23: centre, then for i = 1..4 the +i tap, then the -i tap.
VKR_DEV float blur9(float m4, float m3, float m2, float m1, float c, float p1, float p2, float p3, float p4)
{
    float acc = c * BW0;
    acc = fma_(p1, BW1, acc); acc = fma_(m1, BW1, acc);
    acc = fma_(p2, BW2, acc); acc = fma_(m2, BW2, acc);
    acc = fma_(p3, BW3, acc); acc = fma_(m3, BW3, acc);
    acc = fma_(p4, BW4, acc); acc = fma_(m4, BW4, acc);
    return acc;
}

// Horizontal pass.  One workgroup blurs a 512-pixel piece of one row: the piece plus a 4-pixel
// apron on each side is staged in LDS (each global texel is read once, 16 bytes per lane), then
// every lane produces two adjacent output pixels from five 16-byte LDS reads.
#ifndef VKR_BH_ROWS
#define VKR_BH_ROWS 1
#endif
static constexpr int BH_PIECE = 512;

__global__ __launch_bounds__(256) void bloom_h_kernel(const BloomParams p)
{
    __shared__ uint4 tile[(BH_PIECE + 8) / 2];  // pairs of RGBA16F texels
    const int t = threadIdx.x;
    const int row = blockIdx.y;
    const int x0 = blockIdx.x * BH_PIECE;
    const uint2 *srow = p.src + (size_t)(row + p.src_row_offset) * p.width;
    uint2 *tile2 = reinterpret_cast<uint2 *>(tile);

    // body: pixels x0 + 2t, x0 + 2t + 1 -> tile slots 4 + 2t, 5 + 2t
    const int xa = x0 + 2 * t;
    if (xa + 1 < p.width && (p.width & 1) == 0) {
        tile[2 + t] = *reinterpret_cast<const uint4 *>(srow + xa);
    } else {
        tile2[4 + 2 * t] = srow[wrap_index(xa, p.width, p.wrap, p.clamp_lo, p.clamp_hi)];
        tile2[5 + 2 * t] = srow[wrap_index(xa + 1, p.width, p.wrap, p.clamp_lo, p.clamp_hi)];
    }
    if (t < 8) {  // aprons: x0-4..x0-1 and x0+512..x0+515
        int xs = t < 4 ? x0 - 4 + t : x0 + BH_PIECE + (t - 4);
        int slot = t < 4 ? t : BH_PIECE + 4 + (t - 4);
        tile2[slot] = srow[wrap_index(xs, p.width, p.wrap, p.clamp_lo, p.clamp_hi)];
    }
    __syncthreads();
    if (xa >= p.width) return;

    // outputs 2t and 2t+1 need tile slots 2t .. 2t+9
    float r[10], g[10], b[10];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        uint4 v = tile[t + q];
        rgba lo = unpack4(make_uint2(v.x, v.y)), hi = unpack4(make_uint2(v.z, v.w));
        r[2 * q] = lo.r; g[2 * q] = lo.g; b[2 * q] = lo.b;
        r[2 * q + 1] = hi.r; g[2 * q + 1] = hi.g; b[2 * q + 1] = hi.b;
    }
    uint2 o0 = pack4(blur9(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]),
                     blur9(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8]),
                     blur9(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]), 1.0f);
    uint2 o1 = pack4(blur9(r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]),
                     blur9(g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9]),
                     blur9(b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9]), 1.0f);
    uint2 *drow = p.dst + (size_t)row * p.width;
    if (xa + 1 < p.width && (p.width & 1) == 0) {
        *reinterpret_cast<uint4 *>(drow + xa) = make_uint4(o0.x, o0.y, o1.x, o1.y);
    } else {
        drow[xa] = o0;
        if (xa + 1 < p.width) drow[xa + 1] = o1;
    }
}

// postprocess.frag:11-18
VKR_DEV float aces(float c)
{
    float num = c * fma_(2.51f, c, 0.03f);
    float den = fma_(c, fma_(2.43f, c, 0.59f), 0.14f);  // >= 0.14 for c >= 0: inside fast_rcp_rn's exact domain
    return clamp01(num * (c >= 0.0f ? fast_rcp_rn(den) : rcp_rn(den)));
}

// pow(m, 1/2.2) for m in [0,1] through the hardware log2/exp2 units
VKR_DEV float gamma22(float m) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(m) * (1.0f / 2.2f)); }

VKR_DEV float linear_to_srgb(float x)
{
    return x <= 0.0031308f ? 12.92f * x : fma_(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f)), -0.055f);
}

VKR_DEV uint32_t tonemap_pack(float r, float g, float b, int flags)
{
    float c[3] = {r, g, b};
    uint32_t out = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = gamma22(aces(c[k]));
        if (flags & VKR_POST_SRGB_TARGET) v = linear_to_srgb(clamp01(v));
        uint32_t q = (uint32_t)__builtin_floorf(fma_(clamp01(v), 255.0f, 0.5f));
        out |= q << (8 * k);
    }
    return out;
}

// Vertical pass, optionally fused with the post-process draw.  A workgroup owns 256 * PXL columns
// and VR output rows; each lane walks down its PXL adjacent columns with the nine source rows of the
// window held in registers, so a source row is fetched once per workgroup (plus the 8-row apron).
#ifndef VKR_VR
#define VKR_VR 16
#endif
#ifndef VKR_PXL
#define VKR_PXL 1
#endif
static constexpr int VR = VKR_VR, PXL = VKR_PXL;

template <int N> struct TexelVec;
template <> struct TexelVec<1> { typedef uint2 type; };
template <> struct TexelVec<2> { typedef uint4 type; };

VKR_DEV void unpack_n(uint2 v, rgba (&o)[1]) { o[0] = unpack4(v); }
VKR_DEV void unpack_n(uint4 v, rgba (&o)[2]) { o[0] = unpack4(make_uint2(v.x, v.y)); o[1] = unpack4(make_uint2(v.z, v.w)); }

template <bool FUSE_POST>
__global__ __launch_bounds__(256) void bloom_v_kernel(const BloomParams p, const uint2 *hdr, uint32_t *out8, int flags)
{
    typedef typename TexelVec<PXL>::type vec_t;
    const int x = (blockIdx.x * 256 + threadIdx.x) * PXL;
    if (x >= p.width) return;
    // the vector path needs all PXL columns inside the image and 16-byte aligned rows
    const bool vec_ok = PXL == 1 || (x + PXL <= p.width && (p.width % PXL) == 0);
    const int r0 = blockIdx.y * VR;
    float wr[9][PXL], wg[9][PXL], wb[9][PXL];
    auto load_row = [&](int slot, int src_row) {
        const uint2 *row = p.src + (size_t)src_row * p.width + x;
        rgba t[PXL];
        if (vec_ok) {
            unpack_n(*reinterpret_cast<const vec_t *>(row), t);
        } else {
#pragma unroll
            for (int k = 0; k < PXL; k++) t[k] = unpack4(row[min(k, p.width - 1 - x)]);
        }
#pragma unroll
        for (int k = 0; k < PXL; k++) { wr[slot][k] = t[k].r; wg[slot][k] = t[k].g; wb[slot][k] = t[k].b; }
    };
#pragma unroll
    for (int kk = 0; kk < 8; kk++)
        load_row(kk, wrap_index(r0 + kk - 4 + p.src_row_offset, p.src_rows, p.wrap, p.clamp_lo, p.clamp_hi));
#pragma unroll
    for (int r = 0; r < VR; r++) {
        if (r0 + r >= p.rows) break;
        load_row((r + 8) % 9, wrap_index(r0 + r + 4 + p.src_row_offset, p.src_rows, p.wrap, p.clamp_lo, p.clamp_hi));
        const size_t o = (size_t)(r0 + r) * p.width + x;
        uint32_t px8[PXL];
        uint2 px16[PXL];
        rgba h[PXL];
        if (FUSE_POST) {
            if (vec_ok) {
                unpack_n(*reinterpret_cast<const vec_t *>(hdr + o), h);
            } else {
#pragma unroll
                for (int k = 0; k < PXL; k++) h[k] = unpack4(hdr[o + min(k, p.width - 1 - x)]);
            }
        }
#pragma unroll
        for (int k = 0; k < PXL; k++) {
#define W_(arr, q) arr[(r + (q)) % 9][k]
            float br = blur9(W_(wr, 0), W_(wr, 1), W_(wr, 2), W_(wr, 3), W_(wr, 4), W_(wr, 5), W_(wr, 6), W_(wr, 7), W_(wr, 8));
            float bg = blur9(W_(wg, 0), W_(wg, 1), W_(wg, 2), W_(wg, 3), W_(wg, 4), W_(wg, 5), W_(wg, 6), W_(wg, 7), W_(wg, 8));
            float bb = blur9(W_(wb, 0), W_(wb, 1), W_(wb, 2), W_(wb, 3), W_(wb, 4), W_(wb, 5), W_(wb, 6), W_(wb, 7), W_(wb, 8));
#undef W_
            if (FUSE_POST)  // the bloom-2 attachment is RGBA16F: round before postprocess.frag reads it
                px8[k] = tonemap_pack(h[k].r + round_f16(br), h[k].g + round_f16(bg), h[k].b + round_f16(bb), flags);
            else
                px16[k] = pack4(br, bg, bb, 1.0f);
        }
        if (FUSE_POST) {
            if (vec_ok && PXL == 2) *reinterpret_cast<uint2 *>(out8 + o) = make_uint2(px8[0], px8[PXL - 1]);
            else
#pragma unroll
                for (int k = 0; k < PXL; k++) if (x + k < p.width) out8[o + k] = px8[k];
        } else {
            if (vec_ok && PXL == 2) *reinterpret_cast<uint4 *>(p.dst + o) = make_uint4(px16[0].x, px16[0].y, px16[PXL - 1].x, px16[PXL - 1].y);
            else
#pragma unroll
                for (int k = 0; k < PXL; k++) if (x + k < p.width) p.dst[o + k] = px16[k];
        }
    }
}

// Stand-alone post-process draw: two pixels per lane (16-byte loads, 8-byte store).
__global__ __launch_bounds__(256) void postprocess_kernel(const uint4 *hdr, const uint4 *bloom, uint2 *out, size_t pairs, int flags)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    uint4 h = hdr[i], b = bloom[i];
    rgba h0 = unpack4(make_uint2(h.x, h.y)), h1 = unpack4(make_uint2(h.z, h.w));
    rgba b0 = unpack4(make_uint2(b.x, b.y)), b1 = unpack4(make_uint2(b.z, b.w));
    out[i] = make_uint2(tonemap_pack(h0.r + b0.r, h0.g + b0.g, h0.b + b0.b, flags),
                        tonemap_pack(h1.r + b1.r, h1.g + b1.g, h1.b + b1.b, flags));
}

__global__ __launch_bounds__(256) void postprocess_tail_kernel(const uint2 *hdr, const uint2 *bloom, uint32_t *out, size_t first, size_t count, int flags)
{
    size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    rgba h = unpack4(hdr[i]), b = unpack4(bloom[i]);
    out[i] = tonemap_pack(h.r + b.r, h.g + b.g, h.b + b.b, flags);
}

static int32_t check_bloom(const char *who, const void *src, int width, int rows, int src_rows, int src_row_offset, int horizontal, int wrap, int lo, int hi)
{
    VKR_REQUIRE(src != nullptr, "%s: source image is null", who);
    VKR_REQUIRE(width > 0 && rows > 0 && src_rows > 0, "%s: bad size %dx%d (src_rows %d)", who, width, rows, src_rows);
    VKR_REQUIRE(wrap == VKR_WRAP_REPEAT || wrap == VKR_WRAP_CLAMP, "%s: unknown wrap mode %d", who, wrap);
    int n = horizontal ? width : src_rows;
    if (wrap == VKR_WRAP_CLAMP) VKR_REQUIRE(lo >= 0 && hi < n && lo <= hi, "%s: clamp range [%d,%d] outside [0,%d)", who, lo, hi, n);
    if (horizontal) VKR_REQUIRE(src_row_offset >= 0 && rows + src_row_offset <= src_rows, "%s: rows %d + offset %d exceed src_rows %d", who, rows, src_row_offset, src_rows);
    return VKR_OK;
}

}  // namespace vkr

extern "C" int32_t vkr_bloom_blur(const VkrBloomArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_bloom_blur: args is null");
    VKR_REQUIRE(a->dst != nullptr, "vkr_bloom_blur: destination image is null");
    int32_t rc = check_bloom("vkr_bloom_blur", a->src, a->width, a->rows, a->src_rows, a->src_row_offset, a->horizontal, a->wrap, a->clamp_lo, a->clamp_hi);
    if (rc != VKR_OK) return rc;
    BloomParams p;
    p.src = (const uint2 *)a->src; p.dst = (uint2 *)a->dst;
    p.width = a->width; p.rows = a->rows; p.src_rows = a->src_rows; p.src_row_offset = a->src_row_offset;
    p.wrap = a->wrap; p.clamp_lo = a->clamp_lo; p.clamp_hi = a->clamp_hi;
    if (a->horizontal) {
        dim3 grid((a->width + BH_PIECE - 1) / BH_PIECE, a->rows);
        hipLaunchKernelGGL(bloom_h_kernel, grid, dim3(256), 0, as_stream(stream), p);
    } else {
        dim3 grid((a->width + 256 * PXL - 1) / (256 * PXL), (a->rows + VR - 1) / VR);
        hipLaunchKernelGGL(bloom_v_kernel<false>, grid, dim3(256), 0, as_stream(stream), p, (const uint2 *)nullptr, (uint32_t *)nullptr, 0);
    }
    return launch_status("vkr_bloom_blur");
}

extern "C" int32_t vkr_bloomv_postprocess(const VkrBloomPostArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_bloomv_postprocess: args is null");
    VKR_REQUIRE(a->hdr != nullptr && a->out_rgba8 != nullptr, "vkr_bloomv_postprocess: hdr or output image is null");
    int32_t rc = check_bloom("vkr_bloomv_postprocess", a->bloom1, a->width, a->rows, a->src_rows, a->src_row_offset, 0, a->wrap, a->clamp_lo, a->clamp_hi);
    if (rc != VKR_OK) return rc;
    BloomParams p;
    p.src = (const uint2 *)a->bloom1; p.dst = nullptr;
    p.width = a->width; p.rows = a->rows; p.src_rows = a->src_rows; p.src_row_offset = a->src_row_offset;
    p.wrap = a->wrap; p.clamp_lo = a->clamp_lo; p.clamp_hi = a->clamp_hi;
    dim3 grid((a->width + 256 * PXL - 1) / (256 * PXL), (a->rows + VR - 1) / VR);
    hipLaunchKernelGGL(bloom_v_kernel<true>, grid, dim3(256), 0, as_stream(stream), p, (const uint2 *)a->hdr, (uint32_t *)a->out_rgba8, a->flags);
    return launch_status("vkr_bloomv_postprocess");
}

extern "C" int32_t vkr_postprocess(const void *hdr, const void *bloom, void *out_rgba8, int32_t width, int32_t rows, int32_t flags, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(hdr && bloom && out_rgba8, "vkr_postprocess: an image is null");
    VKR_REQUIRE(width > 0 && rows > 0, "vkr_postprocess: bad size %dx%d", width, rows);
    size_t count = (size_t)width * rows, pairs = count / 2;
    if (pairs) {
        hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, as_stream(stream),
                           (const uint4 *)hdr, (const uint4 *)bloom, (uint2 *)out_rgba8, pairs, flags);
        int32_t rc = launch_status("vkr_postprocess");
        if (rc != VKR_OK) return rc;
    }
    if (count & 1) {
        hipLaunchKernelGGL(postprocess_tail_kernel, dim3(1), dim3(256), 0, as_stream(stream),
                           (const uint2 *)hdr, (const uint2 *)bloom, (uint32_t *)out_rgba8, pairs * 2, count, flags);
        return launch_status("vkr_postprocess");
    }
    return VKR_OK;
}
