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

// Horizontal pass.  A wave loads 128 adjacent pixels (two per lane, 16 bytes), every lane gets the packed texels of
// the lanes two and one to its left and right through DPP wave shifts (sixteen v_mov_b32 with a DPP modifier), and
// lanes 2..61 produce their two pixels; consecutive waves overlap by four pixels on each side (6.7 % of the loads,
// served by the cache).  No LDS and no barrier: measured against staging a 512-pixel piece and its aprons in LDS,
// 21.5 against 22.6 us at 3840x2160 and 97.2 against 100.1 us at 16384x2048, the same bits.
VKR_DEV uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }  // lane i <- lane i - 1
VKR_DEV uint32_t wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }  // lane i <- lane i + 1
VKR_DEV uint4 wave_shr1(uint4 v) { return make_uint4(wave_shr1(v.x), wave_shr1(v.y), wave_shr1(v.z), wave_shr1(v.w)); }
VKR_DEV uint4 wave_shl1(uint4 v) { return make_uint4(wave_shl1(v.x), wave_shl1(v.y), wave_shl1(v.z), wave_shl1(v.w)); }
static constexpr int BH_DPP_OUT = 120;  // output pixels per wave

__global__ __launch_bounds__(256) void bloom_h_kernel(const BloomParams p)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row = blockIdx.y;
    const int base = wave * BH_DPP_OUT - 4;  // first pixel the wave loads
    if (base + 4 >= p.width) return;         // wave-uniform
    const uint2 *srow = p.src + (size_t)(row + p.src_row_offset) * p.width;
    const int xa = base + 2 * lane;
    uint4 v;
    if (xa >= 0 && xa + 1 < p.width && (p.width & 1) == 0) {
        v = *reinterpret_cast<const uint4 *>(srow + xa);
    } else {
        const uint2 a = srow[wrap_index(xa, p.width, p.wrap, p.clamp_lo, p.clamp_hi)], b = srow[wrap_index(xa + 1, p.width, p.wrap, p.clamp_lo, p.clamp_hi)];
        v = make_uint4(a.x, a.y, b.x, b.y);
    }
    const uint4 m1 = wave_shr1(v), m2 = wave_shr1(m1), p1 = wave_shl1(v), p2 = wave_shl1(p1);
    const uint4 five[5] = {m2, m1, v, p1, p2};
    float r[10], g[10], b[10];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        rgba lo = unpack4(make_uint2(five[q].x, five[q].y)), hi = unpack4(make_uint2(five[q].z, five[q].w));
        r[2 * q] = lo.r; g[2 * q] = lo.g; b[2 * q] = lo.b;
        r[2 * q + 1] = hi.r; g[2 * q + 1] = hi.g; b[2 * q + 1] = hi.b;
    }
    if (lane < 2 || lane > 61 || xa >= p.width) return;
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

// postprocess.frag:11-18.  The argument is the sum of two binary16 values, so it is at most 131008 in magnitude,
// infinite or NaN.  The denominator 2.43 c^2 + 0.59 c + 0.14 has no real root: for every finite argument it lies
// in [0.104, 4.2e10], inside fast_rcp_rn's exact domain; an infinite or NaN argument gives NaN on the fast and on
// the IEEE path alike, which the clamp turns into 0.
VKR_DEV float aces(float c)
{
    float num = c * fma_(2.51f, c, 0.03f);
    float den = fma_(c, fma_(2.43f, c, 0.59f), 0.14f);
    return clamp01(num * fast_rcp_rn(den));
}

// pow(m, 1/2.2) for m in [0,1] through the hardware log2/exp2 units
VKR_DEV float gamma22(float m) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(m) * (1.0f / 2.2f)); }

VKR_DEV float linear_to_srgb(float x)
{
    return x <= 0.0031308f ? 12.92f * x : fma_(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f)), -0.055f);
}

// One channel to its 8-bit code: floor(clamp(v) * 255 + 0.5).  The product is at least 0.5, so the conversion's
// truncation is the floor.
VKR_DEV uint32_t unorm8(float v) { return (uint32_t)fma_(clamp01(v), 255.0f, 0.5f); }

VKR_DEV uint32_t tonemap_pack(float r, float g, float b, int flags)
{
    float c[3] = {r, g, b};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = gamma22(aces(c[k]));
        if (flags & VKR_POST_SRGB_TARGET) v = linear_to_srgb(clamp01(v));
        q[k] = unorm8(v);
    }
    return (q[0] | 0xFF000000u) | (q[1] << 8) | (q[2] << 16);
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

// Both bloom passes and the post-process draw in one kernel: 8 bytes of the thresholded image and 8 bytes of
// the colour image in, 4 bytes out per pixel; neither bloom image is stored.  A workgroup is ONE wave that owns
// 128 columns (two adjacent pixels per lane, 16-byte loads) and walks down `band` output rows.  Per source row
// the wave stages its 136 texels (4-texel aprons) in LDS, every lane blurs its two pixels horizontally and
// rounds them to binary16 -- the value the bloom-1 attachment would hold (bloom_blur.frag:14-33 through an
// RGBA16F target) -- into a nine-row register window; the vertical blur, the rounding of the bloom-2 attachment,
// the sum with the colour image and postprocess.frag:24-38 follow from registers.  Loads run BF_PF rows ahead of
// their use.  With one wave per workgroup the LDS hand-over needs no barrier between waves.
// The kernel is bound by VALU issue, not by memory (about 280 vector instructions per row step of 128 pixels:
// 54 fmas for each blur, 13 per colour channel for ACES, gamma and the 8-bit code, conversions): what pays is
// occupancy (four waves per SIMD: 43 us at 3840x2160 against 60 with three) and a row loop without per-lane
// branches; a deeper load pipeline (BF_PF 3) costs registers and is slower.
#ifndef VKR_BF_PF
#define VKR_BF_PF 1
#endif
#ifndef VKR_BF_WAVES
#define VKR_BF_WAVES 4
#endif
static constexpr int BF_COLS = 128, BF_PF = VKR_BF_PF;
static_assert(9 % BF_PF == 0, "the prefetch slots rotate with the nine-row window");

struct BloomFusedParams {
    const uint2 *thr, *hdr; uint32_t *out8;
    int width, rows, src_rows, src_row_offset;
    int wrap_h, lo_h, hi_h, wrap_v, lo_v, hi_v;
    int band;  // output rows per workgroup
    int flags;
};

struct px2 { float r0, g0, b0, r1, g1, b1; };  // the two pixels of a lane

// float(one binary16 half of `pair`) + float(b) in one mixed-precision fma: x * 1 + y is the correctly rounded sum,
// and the two conversions to binary32 and the add (slow-class instructions, three of them) become one.
template <int HI> VKR_DEV float add_half(uint32_t pair, _Float16 b)
{
    float o;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(o) : "v"(pair), "v"(b));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(o) : "v"(pair), "v"(b));
    return o;
}

// EVEN_REPEAT: the width is even and pass 1 wraps with REPEAT (the reference's state).  Every lane's pair of
// columns is then an aligned 16-byte load, also where a tile reaches over the right edge (the pair wraps as a
// pair), and the row loop has no per-lane branches.
template <bool SRGB, bool EVEN_REPEAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VKR_BF_WAVES, VKR_BF_WAVES))) void bloom_post_fused_kernel(const BloomFusedParams p)
{
    constexpr int flags = SRGB ? VKR_POST_SRGB_TARGET : 0;
    __shared__ __align__(16) float tile[2][3][BF_COLS + 8];  // binary32 r, g, b planes of one source row, two rows in turn
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * BF_COLS;
    const int xa = x0 + 2 * t;
    const int r0 = blockIdx.y * p.band;
    const int nrows = min(p.band, p.rows - r0);
    const bool inside = xa + 1 < p.width;
    const bool vec = EVEN_REPEAT || ((p.width & 1) == 0 && inside);  // this lane's pair is a 16-byte aligned load
    const int xw0 = wrap_index(xa, p.width, p.wrap_h, p.lo_h, p.hi_h), xw1 = wrap_index(xa + 1, p.width, p.wrap_h, p.lo_h, p.hi_h);
    const int xh = EVEN_REPEAT ? min(xa, p.width - 2) : xa;  // colour-image column (lanes past the right edge load a valid pair and drop it)
    const int apron_x = wrap_index(t < 4 ? x0 - 4 + t : x0 + BF_COLS + (t - 4), p.width, p.wrap_h, p.lo_h, p.hi_h);
    const int apron_slot = t < 4 ? t : BF_COLS + 4 + (t - 4);

    uint4 pf_body[BF_PF], pf_hdr[BF_PF];
    uint2 pf_apron[BF_PF];
    auto fetch = [&](int slot, int k) {  // source row k of the band (k = 0 is four rows above its first output row)
        if (k < nrows + 8) {
            int sr = wrap_index(r0 + k - 4 + p.src_row_offset, p.src_rows, p.wrap_v, p.lo_v, p.hi_v);
            const uint2 *row = p.thr + (size_t)sr * p.width;
            if (vec) pf_body[slot] = *reinterpret_cast<const uint4 *>(row + (EVEN_REPEAT ? xw0 : xa));
            else { uint2 a = row[xw0], b = row[xw1]; pf_body[slot] = make_uint4(a.x, a.y, b.x, b.y); }
            if (t < 8) pf_apron[slot] = row[apron_x];
        }
        int r = k - 8;  // the output row that source row completes
        if (r >= 0 && r < nrows) {
            const uint2 *row = p.hdr + (size_t)(r0 + r) * p.width;
            if (vec) pf_hdr[slot] = *reinterpret_cast<const uint4 *>(row + xh);
            else { uint2 a = row[min(xa, p.width - 1)], b = row[min(xa + 1, p.width - 1)]; pf_hdr[slot] = make_uint4(a.x, a.y, b.x, b.y); }
        }
    };
    px2 win[9];  // the bloom-1 values (binary16) as binary32: a row is converted once, not once per output row it feeds
    auto hblur = [&](int slot, int wslot, int k) {
        // Every lane converts its own two texels (and eight lanes an apron texel) to binary32 and shares them
        // through LDS: six conversions per lane and row instead of thirty, and a blur of full-rate fmas.
        float (*buf)[BF_COLS + 8] = tile[k & 1];
        const uint4 v = pf_body[slot];
        const rgba lo = unpack4(make_uint2(v.x, v.y)), hi = unpack4(make_uint2(v.z, v.w));
        *reinterpret_cast<float2 *>(&buf[0][4 + 2 * t]) = make_float2(lo.r, hi.r);
        *reinterpret_cast<float2 *>(&buf[1][4 + 2 * t]) = make_float2(lo.g, hi.g);
        *reinterpret_cast<float2 *>(&buf[2][4 + 2 * t]) = make_float2(lo.b, hi.b);
        if (t < 8) {
            const rgba a = unpack4(pf_apron[slot]);
            buf[0][apron_slot] = a.r; buf[1][apron_slot] = a.g; buf[2][apron_slot] = a.b;
        }
        __syncthreads();
        float c[3][10];
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const float2 f = *reinterpret_cast<const float2 *>(&buf[ch][2 * t + 2 * q]);
                c[ch][2 * q] = f.x; c[ch][2 * q + 1] = f.y;
            }
        const float *r = c[0], *g = c[1], *b = c[2];
        px2 o;
        o.r0 = pin(round_f16(blur9(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8])));
        o.g0 = pin(round_f16(blur9(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8])));
        o.b0 = pin(round_f16(blur9(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8])));
        o.r1 = pin(round_f16(blur9(r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9])));
        o.g1 = pin(round_f16(blur9(g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9])));
        o.b1 = pin(round_f16(blur9(b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9])));
        win[wslot] = o;
    };

#pragma unroll
    for (int k = 0; k < BF_PF; k++) fetch(k, k);
#pragma unroll
    for (int k = 0; k < 8; k++) {  // rows above the first output row
        hblur(k % BF_PF, k, k);
        fetch(k % BF_PF, k + BF_PF);
    }
    for (int rb = 0; rb < nrows; rb += 9) {
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int r = rb + j, k = r + 8;
            if (r >= nrows) return;
            hblur((j + 8) % BF_PF, (j + 8) % 9, k);
            const uint4 hv = pf_hdr[(j + 8) % BF_PF];
            fetch((j + 8) % BF_PF, k + BF_PF);
#define W_(ch, q) win[(j + (q)) % 9].ch
            float br0 = blur9(W_(r0, 0), W_(r0, 1), W_(r0, 2), W_(r0, 3), W_(r0, 4), W_(r0, 5), W_(r0, 6), W_(r0, 7), W_(r0, 8));
            float bg0 = blur9(W_(g0, 0), W_(g0, 1), W_(g0, 2), W_(g0, 3), W_(g0, 4), W_(g0, 5), W_(g0, 6), W_(g0, 7), W_(g0, 8));
            float bb0 = blur9(W_(b0, 0), W_(b0, 1), W_(b0, 2), W_(b0, 3), W_(b0, 4), W_(b0, 5), W_(b0, 6), W_(b0, 7), W_(b0, 8));
            float br1 = blur9(W_(r1, 0), W_(r1, 1), W_(r1, 2), W_(r1, 3), W_(r1, 4), W_(r1, 5), W_(r1, 6), W_(r1, 7), W_(r1, 8));
            float bg1 = blur9(W_(g1, 0), W_(g1, 1), W_(g1, 2), W_(g1, 3), W_(g1, 4), W_(g1, 5), W_(g1, 6), W_(g1, 7), W_(g1, 8));
            float bb1 = blur9(W_(b1, 0), W_(b1, 1), W_(b1, 2), W_(b1, 3), W_(b1, 4), W_(b1, 5), W_(b1, 6), W_(b1, 7), W_(b1, 8));
#undef W_
            // colour + bloom 2 (postprocess.frag:27-29), both read as binary16
            uint32_t o0 = tonemap_pack(add_half<0>(hv.x, to_f16(br0)), add_half<1>(hv.x, to_f16(bg0)), add_half<0>(hv.y, to_f16(bb0)), flags);
            uint32_t o1 = tonemap_pack(add_half<0>(hv.z, to_f16(br1)), add_half<1>(hv.z, to_f16(bg1)), add_half<0>(hv.w, to_f16(bb1)), flags);
            uint32_t *orow = p.out8 + (size_t)(r0 + r) * p.width;
            if (EVEN_REPEAT) { if (inside) *reinterpret_cast<uint2 *>(orow + xa) = make_uint2(o0, o1); }
            else if (vec) *reinterpret_cast<uint2 *>(orow + xa) = make_uint2(o0, o1);
            else {
                if (xa < p.width) orow[xa] = o0;
                if (xa + 1 < p.width) orow[xa + 1] = o1;
            }
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
        dim3 grid((a->width + 4 * BH_DPP_OUT - 1) / (4 * BH_DPP_OUT), a->rows);
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

extern "C" int32_t vkr_bloom_postprocess(const VkrBloomChainArgs *a, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(a != nullptr, "vkr_bloom_postprocess: args is null");
    VKR_REQUIRE(a->threshold != nullptr && a->hdr != nullptr && a->out_rgba8 != nullptr, "vkr_bloom_postprocess: an image is null");
    VKR_REQUIRE(a->width > 0 && a->rows > 0 && a->src_rows > 0, "vkr_bloom_postprocess: bad size %dx%d (src_rows %d)", a->width, a->rows, a->src_rows);
    VKR_REQUIRE((a->wrap_h == VKR_WRAP_REPEAT || a->wrap_h == VKR_WRAP_CLAMP) && (a->wrap_v == VKR_WRAP_REPEAT || a->wrap_v == VKR_WRAP_CLAMP),
                "vkr_bloom_postprocess: unknown wrap mode %d / %d", a->wrap_h, a->wrap_v);
    if (a->wrap_h == VKR_WRAP_CLAMP)
        VKR_REQUIRE(a->clamp_lo_h >= 0 && a->clamp_hi_h < a->width && a->clamp_lo_h <= a->clamp_hi_h,
                    "vkr_bloom_postprocess: column clamp range [%d,%d] outside [0,%d)", a->clamp_lo_h, a->clamp_hi_h, a->width);
    if (a->wrap_v == VKR_WRAP_CLAMP)
        VKR_REQUIRE(a->clamp_lo_v >= 0 && a->clamp_hi_v < a->src_rows && a->clamp_lo_v <= a->clamp_hi_v,
                    "vkr_bloom_postprocess: row clamp range [%d,%d] outside [0,%d)", a->clamp_lo_v, a->clamp_hi_v, a->src_rows);
    BloomFusedParams p;
    p.thr = (const uint2 *)a->threshold; p.hdr = (const uint2 *)a->hdr; p.out8 = (uint32_t *)a->out_rgba8;
    p.width = a->width; p.rows = a->rows; p.src_rows = a->src_rows; p.src_row_offset = a->src_row_offset;
    p.wrap_h = a->wrap_h; p.lo_h = a->clamp_lo_h; p.hi_h = a->clamp_hi_h;
    p.wrap_v = a->wrap_v; p.lo_v = a->clamp_lo_v; p.hi_v = a->clamp_hi_v;
    p.flags = a->flags;
    // rows per wave: long bands blur fewer apron rows (8 per band), short ones give the chip more waves; aim at
    // sixteen waves per CU (one round of four per SIMD) and stay between 16 rows (apron 50 %) and 32 (25 %)
    const int tiles_x = (a->width + BF_COLS - 1) / BF_COLS;
    long band = ((long)a->rows * tiles_x + 4095) / 4096;
    p.band = (int)(band < 16 ? 16 : band > 32 ? 32 : band);
#ifdef VKR_EXPERIMENT_KNOBS  // kernel experiments only: the shipped library reads no environment
    static const int forced_band = getenv("VKR_BF_BAND") ? atoi(getenv("VKR_BF_BAND")) : 0;
    if (forced_band > 0) p.band = forced_band;
#endif
    dim3 grid(tiles_x, (a->rows + p.band - 1) / p.band);
    const bool srgb = (a->flags & VKR_POST_SRGB_TARGET) != 0;
    const bool even_repeat = (a->width & 1) == 0 && a->wrap_h == VKR_WRAP_REPEAT;
    auto kernel = srgb ? (even_repeat ? bloom_post_fused_kernel<true, true> : bloom_post_fused_kernel<true, false>)
                       : (even_repeat ? bloom_post_fused_kernel<false, true> : bloom_post_fused_kernel<false, false>);
    hipLaunchKernelGGL(kernel, grid, dim3(64), 0, as_stream(stream), p);
    return launch_status("vkr_bloom_postprocess");
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
