// IBL precompute on gfx950: equirectangular -> cube, irradiance convolution, GGX prefilter and
// the split-sum BRDF LUT (the four Create* functions of VulkanRenderer.cs:653-989).
//
// sin/cos of the fixed sample angles (Hammersley phi_i, the irradiance loop's phi/theta values)
// and the sRGB decode table come from the host C library and are uploaded once per device, so
// kernels never evaluate a transcendental function in the sample loops.
#include <math.h>
#include <mutex>
#include "vkr_device.h"
#include "vkr_host.h"
#include "vkr_irr_bounds.h"

namespace vkr {

static constexpr float PI_F = 3.14159265359f;
static constexpr int SAMPLE_COUNT = 1024;       // prefilter.frag:29, brdf.frag:93
static constexpr int IRR_MAX_PHI = 256, IRR_MAX_THETA = 64;

__device__ float2 g_hammersley_cs[SAMPLE_COUNT];  // (cos, sin) of phi_i = 2*PI * i/1024
__device__ float2 g_irr_phi_cs[IRR_MAX_PHI];      // (cos, sin) of the irradiance loop's phi values
__device__ float2 g_irr_theta_sc[IRR_MAX_THETA];  // (sin, cos) of its theta values
__device__ float g_srgb_lut[256];                 // sRGB 8-bit -> linear

static int g_irr_nphi = 0, g_irr_ntheta = 0;
static std::mutex g_init_mutex;
static bool g_device_ready[64];

// Uploads the tables to the current device on first use.
static int32_t ensure_tables()
{
    int dev = 0;
    VKR_HIP_CHECK(hipGetDevice(&dev));
    VKR_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (g_device_ready[dev]) return VKR_OK;

    static float2 ham[SAMPLE_COUNT], phi_cs[IRR_MAX_PHI], theta_sc[IRR_MAX_THETA];
    static float srgb[256];
    for (int i = 0; i < SAMPLE_COUNT; i++) {
        float phi = (2.0f * PI_F) * ((float)i * (1.0f / 1024.0f));  // prefilter.frag:84
        ham[i] = make_float2(cosf(phi), sinf(phi));
    }
    // irradiance.frag:21-26: the loop variables are binary32 running sums
    int nphi = 0, ntheta = 0;
    for (float phi = 0.0f; phi < 2.0f * PI_F; phi += 0.025f) {
        VKR_REQUIRE(nphi < IRR_MAX_PHI, "irradiance phi table overflow");
        phi_cs[nphi++] = make_float2(cosf(phi), sinf(phi));
    }
    for (float th = 0.0f; th < 0.5f * PI_F; th += 0.025f) {
        VKR_REQUIRE(ntheta < IRR_MAX_THETA, "irradiance theta table overflow");
        theta_sc[ntheta++] = make_float2(sinf(th), cosf(th));
    }
    g_irr_nphi = nphi; g_irr_ntheta = ntheta;
    for (int i = 0; i < 256; i++) {
        float c = (float)i * (1.0f / 255.0f);
        srgb[i] = c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f);
    }
    VKR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_hammersley_cs), ham, sizeof(ham)));
    VKR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_irr_phi_cs), phi_cs, sizeof(phi_cs)));
    VKR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_irr_theta_sc), theta_sc, sizeof(theta_sc)));
    VKR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_srgb_lut), srgb, sizeof(srgb)));
    g_device_ready[dev] = true;
    return VKR_OK;
}

// ---- equirectangular_to_cubemap.frag:8-24 ---------------------------------------------------

VKR_DEV f3 fetch_src(const void *src, int fmt, int w, int x, int y)
{
    size_t t = (size_t)y * w + x;
    if (fmt == VKR_SRC_RGBA8_SRGB) {
        uint32_t v = ((const uint32_t *)src)[t];
        return mk3(g_srgb_lut[v & 0xFF], g_srgb_lut[(v >> 8) & 0xFF], g_srgb_lut[(v >> 16) & 0xFF]);
    }
    rgba c = load_rgba16f(src, t);
    return mk3(c.r, c.g, c.b);
}

// atan2 to about 2.5e-7 rad (the error of a correctly rounded result near pi is 1.2e-7): atan(q) on [0, 1] as q P(q^2),
// P of degree 8 fitted at Chebyshev nodes (1.0e-7), then the octant and the signs.  atan2(0, 0) = 0 like the C
// library's.  Twenty-odd instructions where the device library's atan2f takes three times as many: the pass was
// bound by them, not by the 8 bytes per texel it writes.
VKR_DEV float atan2_short(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float q = mx > 0.0f ? mn * fast_rcp_rn(mx) : 0.0f;
    const float t = q * q;
    float p = fma_(t, 0.0028340642f, -0.01600503f);
    p = fma_(t, p, 0.042587608f);
    p = fma_(t, p, -0.07495446f);
    p = fma_(t, p, 0.10636754f);
    p = fma_(t, p, -0.14202571f);
    p = fma_(t, p, 0.19992484f);
    p = fma_(t, p, -0.33333066f);
    p = fma_(t, p, 1.0f) * q;
    p = ay > ax ? 1.57079637f - p : p;
    p = x < 0.0f ? 3.14159274f - p : p;
    return __builtin_copysignf(p, y);
}

// rn = 1 / n as the host's division rounds it (what face_dir computes per texel); the launch has one row of the cube
// per blockIdx.y.
__global__ __launch_bounds__(256) void equirect_kernel(const void *src, int sw, int sh, int fmt, uint2 *cube, int n, float rn, int row_begin, int row_end)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int row = row_begin + blockIdx.y;
    if (i >= n || row >= row_end) return;
    const int f = (int)(((float)row + 0.5f) * rn);  // row / n: exact, rows stay far below 2^22
    const int j = row - f * n;
    const float s = fma_((float)(2 * i + 1), rn, -1.0f), t = fma_((float)(2 * j + 1), rn, -1.0f);  // face_dir
    const f3 d = normalize3_fast(face_point(f, s, t));
    // asin(d.y) as the angle between the direction and its horizontal part (the direction has unit length)
    const float u = fma_(atan2_short(d.z, d.x), 0.1591f, 0.5f);
    const float v = 1.0f - fma_(atan2_short(d.y, fast_sqrt_rn(fma_(d.x, d.x, d.z * d.z))), 0.3183f, 0.5f);
    const float x = u * (float)sw - 0.5f, y = v * (float)sh - 0.5f;
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    const float al = x - fx, be = y - fy;
    // u and v lie in [0, 1], so the four texel coordinates are in [-1, size]: REPEAT is one step; the clamp that follows
    // does nothing for them and keeps anything else (a NaN converts to 0) inside the image
    const int ix = (int)fx, iy = (int)fy;
    const int i0 = min(max(ix < 0 ? ix + sw : ix, 0), sw - 1), i1 = min(max(ix + 1 >= sw ? ix + 1 - sw : ix + 1, 0), sw - 1);
    const int j0 = min(max(iy < 0 ? iy + sh : iy, 0), sh - 1), j1 = min(max(iy + 1 >= sh ? iy + 1 - sh : iy + 1, 0), sh - 1);
    const float w00 = (1.0f - al) * (1.0f - be), w10 = al * (1.0f - be), w01 = (1.0f - al) * be, w11 = al * be;
    const f3 a = fetch_src(src, fmt, sw, i0, j0), b = fetch_src(src, fmt, sw, i1, j0);
    const f3 c = fetch_src(src, fmt, sw, i0, j1), e = fetch_src(src, fmt, sw, i1, j1);
    cube[(size_t)row * n + i] = pack4(fma_(w11, e.x, fma_(w01, c.x, fma_(w10, b.x, w00 * a.x))),
                                      fma_(w11, e.y, fma_(w01, c.y, fma_(w10, b.y, w00 * a.y))),
                                      fma_(w11, e.z, fma_(w01, c.z, fma_(w10, b.z, w00 * a.z))), 1.0f);
}

// ---- irradiance.frag:10-39 ------------------------------------------------------------------

// The shader's own order (one running sum, phi outer, theta inner), one workgroup per output
// texel.  The 15 876 terms are produced 16 phi slices at a time by all lanes (lane = one theta of
// one slice) into LDS; one lane per colour channel then adds the chunk to the running sum in
// sample order while the other waves already produce the next chunk into the second buffer.
static constexpr int IRR_CHUNK_SLICES = 16;
template <bool TILED>
__global__ __launch_bounds__(256) void irradiance_texel_kernel(const void *env, int env_n, uint2 *out, int n, int nphi, int ntheta, int row_begin)
{
    __shared__ float term[2][3][IRR_CHUNK_SLICES * IRR_MAX_THETA];
    __shared__ float sin_theta[IRR_MAX_THETA];
    __shared__ float result[3];
    const int tid = threadIdx.x;
    const int b = tid & 63, w = tid >> 6;
    const int i = blockIdx.x;
    const int row = row_begin + blockIdx.y;
    const int f = row / n, j = row - f * n;
    const f3 N = normalize3(face_dir(f, i, j, n));
    const f3 right = normalize3(mk3(N.z, 0.0f, -N.x));  // normalize(cross((0,1,0), N))
    const f3 up = normalize3(cross3(N, right));
    const float2 tsc = g_irr_theta_sc[b];
    const float st = tsc.x, ct = tsc.y;
    if (tid < ntheta) sin_theta[tid] = g_irr_theta_sc[tid].x;
    float acc = 0.0f;
    for (int a0 = 0, k = 0; a0 < nphi; a0 += IRR_CHUNK_SLICES, k++) {
        float (*buf)[IRR_CHUNK_SLICES * IRR_MAX_THETA] = term[k & 1];
#pragma unroll
        for (int m = 0; m < IRR_CHUNK_SLICES / 4; m++) {
            const int al = w + 4 * m, a = a0 + al;
            if (a < nphi && b < ntheta) {
                const float2 pcs = g_irr_phi_cs[a];
                const float tx = st * pcs.x, ty = st * pcs.y, tz = ct;
                const f3 sv = mk3(fma_(tz, N.x, fma_(ty, up.x, tx * right.x)),
                                  fma_(tz, N.y, fma_(ty, up.y, tx * right.y)),
                                  fma_(tz, N.z, fma_(ty, up.z, tx * right.z)));
                const f3 c = texcube_rgb_linear<TILED>(env, env_n, sv);
                const int p = al * ntheta + b;
                buf[0][p] = c.x * ct; buf[1][p] = c.y * ct; buf[2][p] = c.z * ct;
            }
        }
        __syncthreads();
        if (tid < 3) {
            // the running sum is one dependent fma per term; the terms are read eight ahead so
            // that a step costs the fma's latency, not an LDS round trip
            const int terms = min(IRR_CHUNK_SLICES, nphi - a0) * ntheta;
            const float *mine = buf[tid];
            int q = 0, bb = 0;
            for (; q + 8 <= terms; q += 8) {
                float v[8], st8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    v[u] = mine[q + u];
                    int t = bb + u;
                    t = t >= ntheta ? t - ntheta : t;
                    st8[u] = sin_theta[t];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) acc = fma_(v[u], st8[u], acc);
                bb += 8;
                bb = bb >= ntheta ? bb - ntheta : bb;
            }
            for (; q < terms; q++) {
                acc = fma_(mine[q], sin_theta[bb], acc);
                bb = bb + 1 == ntheta ? 0 : bb + 1;
            }
        }
    }
    if (tid < 3) result[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        const float inv_count = rcp_rn((float)(nphi * ntheta));
        out[(size_t)row * n + i] = pack4((PI_F * result[0]) * inv_count, (PI_F * result[1]) * inv_count, (PI_F * result[2]) * inv_count, 1.0f);
    }
}

// Phi-slice order, one workgroup per output texel: lane a owns phi slice a (252 of the 256 lanes),
// sums its 63 theta samples from zero, and the slice sums are then added in phi order by one lane
// per colour channel.
template <bool TILED>
__global__ __launch_bounds__(256) void irradiance_slices_kernel(const void *env, int env_n, uint2 *out, int n, int nphi, int ntheta, int row_begin)
{
    __shared__ float part[3][IRR_MAX_PHI];
    const int a = threadIdx.x;
    const int i = blockIdx.x;
    const int row = row_begin + blockIdx.y;
    const int f = row / n, j = row - f * n;
    const f3 N = normalize3(face_dir(f, i, j, n));
    const f3 right = normalize3(mk3(N.z, 0.0f, -N.x));  // normalize(cross((0,1,0), N))
    const f3 up = normalize3(cross3(N, right));
    if (a < nphi) {
        const float2 pcs = g_irr_phi_cs[a];
        float pr = 0.0f, pg = 0.0f, pb = 0.0f;
        for (int b = 0; b < ntheta; b++) {
            const float2 tsc = g_irr_theta_sc[b];
            const float st = tsc.x, ct = tsc.y;
            const float tx = st * pcs.x, ty = st * pcs.y, tz = ct;
            const f3 sv = mk3(fma_(tz, N.x, fma_(ty, up.x, tx * right.x)),
                              fma_(tz, N.y, fma_(ty, up.y, tx * right.y)),
                              fma_(tz, N.z, fma_(ty, up.z, tx * right.z)));
            const f3 c = texcube_rgb_linear<TILED>(env, env_n, sv);
            pr = fma_(c.x * ct, st, pr);
            pg = fma_(c.y * ct, st, pg);
            pb = fma_(c.z * ct, st, pb);
        }
        part[0][a] = pr; part[1][a] = pg; part[2][a] = pb;
    }
    __syncthreads();
    if (a < 3) {
        float acc = 0.0f;
        for (int k = 0; k < nphi; k++) acc += part[a][k];
        part[a][0] = acc;
    }
    __syncthreads();
    if (a == 0) {
        const float inv_count = rcp_rn((float)(nphi * ntheta));
        out[(size_t)row * n + i] = pack4((PI_F * part[0][0]) * inv_count, (PI_F * part[1][0]) * inv_count, (PI_F * part[2][0]) * inv_count, 1.0f);
    }
}

// ---- irradiance.frag:10-39, summed by region of the environment -------------------------------
//
// A sample of an output texel lands 32 environment texels from the next one (at 2048^2 faces), so a texel's own
// samples share no cache lines and the kernels above miss L2 more than once per lookup.  Here the environment is cut
// into regions of 512^2 texels (2 MiB, half of one XCD's L2) and the work is ordered by region: a workgroup takes one
// region and IRR_GROUP output texels, finds for each texel the samples whose footprint starts in the region, fetches
// them (from L2: every workgroup resident on an XCD reads the same region) and adds them up.  The sum of a (texel,
// region) pair is defined so that a wave can form it without sorting anything: the pair's k-th sample in the
// shader's order (phi outer, theta inner) goes to accumulator k mod 64 (lane k mod 64), and the 64 accumulators are
// combined pairwise with the partners 32, 16, 8, 4, 2, 1 lanes away.  irradiance_regions_sum_kernel then adds a
// texel's region sums in region order.  The result differs from the single running sum in rounding only.
//
// Which samples of texel N land in a region: in N's tangent frame the region spans a range of polar angles and a
// range of azimuths, found from its corners and edges; samples outside that box of (phi, theta) indices (widened by
// 0.004 rad) cannot land in it.  Every cell of the box is then decided exactly: the sample's direction and footprint
// are computed with the arithmetic of the other kernels, and the sample belongs to the region that holds the
// footprint's base texel (clamped to the face).  A sample has one such region, so it is counted exactly once; the
// count of samples found travels with the sums, and a texel whose count is not nphi * ntheta is reported.
static constexpr int IRR_GROUP = 64;         // output texels per workgroup
static constexpr int IRR_QUEUE = 128;        // entries of a wave's queue of samples to fetch (a ring; at most 63 + 64 are held)
static constexpr int IRR_REGION_SHIFT = 9;   // regions of 512 x 512 texels

// r in the coordinates of face f: (sc, tc) as cube_face gives them on that face, and the major axis signed so that it
// is positive on the face
VKR_DEV f3 to_face_axes(int f, f3 r)
{
    switch (f) {
    case 0: return mk3(-r.z, -r.y, r.x);
    case 1: return mk3(r.z, -r.y, -r.x);
    case 2: return mk3(r.x, r.z, r.y);
    case 3: return mk3(r.x, -r.z, -r.y);
    case 4: return mk3(r.x, -r.y, r.z);
    default: return mk3(-r.x, -r.y, -r.z);
    }
}

// texcube_fetch_footprint for a footprint whose face is the same for the whole wave.  A texel of the footprint that
// lies one step outside the face is taken through edge_map: outside edge e (i = -1, i = n, j = -1, j = n), at
// coordinate t along the edge, lies texel (m[1] + m[2] t, m[3] + m[4] t) of face m[0] -- what cube_fold gives,
// tabulated, because a single such lane makes its whole wave walk whatever code it needs.
template <bool TILED>
VKR_DEV f3 fetch_footprint_on_face(const void *cube, int n, int face, int i0, int j0, float al, float be, const int (*edge_map)[5])
{
    // where the four texels are (32-bit byte offsets from the cube's base, which is the same for the whole wave: the
    // caller keeps cubes below 4 GiB), then the four loads back to back, then the corner's mean
    unsigned at[4];
    bool whole = true;
    unsigned missing = 4;
    if (i0 >= 0 && j0 >= 0 && i0 + 1 < n && j0 + 1 < n) {  // all four on the face
        const unsigned face0 = (unsigned)face * (unsigned)n * (unsigned)n * 8u;
        if (!TILED) {
            at[0] = face0 + (__umul24((unsigned)j0, (unsigned)n) + (unsigned)i0) * 8u; at[1] = at[0] + 8u; at[2] = at[0] + (unsigned)n * 8u; at[3] = at[2] + 8u;
        } else {
            const unsigned tile_row = (unsigned)n * 32u;  // bytes in one row of tiles
            const unsigned c0 = ((unsigned)i0 >> 2) * 128u + ((unsigned)i0 & 3u) * 8u, c1 = ((unsigned)(i0 + 1) >> 2) * 128u + ((unsigned)(i0 + 1) & 3u) * 8u;
            const unsigned r0 = face0 + __umul24((unsigned)j0 >> 2, tile_row) + ((unsigned)j0 & 3u) * 32u;
            const unsigned r1 = face0 + __umul24((unsigned)(j0 + 1) >> 2, tile_row) + ((unsigned)(j0 + 1) & 3u) * 32u;
            at[0] = r0 + c0; at[1] = r0 + c1; at[2] = r1 + c0; at[3] = r1 + c1;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + (k & 1), j = j0 + (k >> 1);
            const bool x_out = i < 0 || i >= n, y_out = j < 0 || j >= n;
            int f = face, ii = i, jj = j;
            if (x_out && y_out) {  // beyond a corner of the cube: no such texel (some texel is loaded and not used)
                missing = k; whole = false;
                ii = i0 + 1 - (k & 1); jj = j0 + 1 - (k >> 1);
            } else if (x_out || y_out) {
                const int *m = edge_map[x_out ? (i < 0 ? 0 : 1) : (j < 0 ? 2 : 3)];
                const int t = x_out ? j : i;
                f = m[0]; ii = m[1] + m[2] * t; jj = m[3] + m[4] * t;
            }
            at[k] = (unsigned)cube_texel_index<TILED>(n, f, ii, jj) * 8u;
        }
    }
    const char *base = (const char *)cube;
    rgba tx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) tx[k] = unpack4(*(const uint2 *)(base + at[k]));
    if (!whole) {  // cube corner: the missing texel is the mean of the other three
        float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) if ((unsigned)k != missing) { sr += tx[k].r; sg += tx[k].g; sb += tx[k].b; }
#pragma unroll
        for (int k = 0; k < 4; k++) if ((unsigned)k == missing) { tx[k].r = sr * (1.0f / 3.0f); tx[k].g = sg * (1.0f / 3.0f); tx[k].b = sb * (1.0f / 3.0f); }
    }
    const float w0 = (1.0f - al) * (1.0f - be), w1 = al * (1.0f - be), w2 = (1.0f - al) * be, w3 = al * be;
    return mk3(fma_(w3, tx[3].r, fma_(w2, tx[2].r, fma_(w1, tx[1].r, w0 * tx[0].r))),
               fma_(w3, tx[3].g, fma_(w2, tx[2].g, fma_(w1, tx[1].g, w0 * tx[0].g))),
               fma_(w3, tx[3].b, fma_(w2, tx[2].b, fma_(w1, tx[1].b, w0 * tx[0].b))));
}

// partial: [region][4][texels_padded] -- red, green, blue sums and the number of samples, texels_padded = n_groups * IRR_GROUP
template <bool TILED>
__global__ __launch_bounds__(256) void irradiance_region_kernel(const void *env, int env_n, int n, int nphi, int ntheta, int row_begin, int n_texels,
                                                                float *partial, int regions_per_side, int n_regions, int n_groups)
{
    __shared__ float2 phi_cs[IRR_MAX_PHI], theta_sc[IRR_MAX_THETA];
    __shared__ float frame[IRR_GROUP][9];
    __shared__ irr_box box[IRR_GROUP];
    __shared__ uint4 queue[4][IRR_QUEUE];  // base texel (+1, 16 bits each), the two weights' bits, theta index
    __shared__ float sums[4][IRR_GROUP];
    __shared__ int edge_map[4][5];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // workgroups go to the XCDs round-robin by id: XCD x works through regions x, x + 8, ... one after the other
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int turn = (k / n_groups) * 8 + xcd, group = k % n_groups;
    if (turn >= n_regions) return;
    // Regions near the middle of a face span 2.5 times the solid angle of those in its corners, and so hold as many
    // times the samples.  With an even number of regions per side the four mirror images of a region on the six faces
    // are 24 regions of equal weight: taken in that order, every XCD gets three of each kind.
    int region = turn;
    if ((regions_per_side & 1) == 0) {
        const int half = regions_per_side >> 1, quad = turn / 24, image = turn % 24, mirror = image / 6;
        const int qx = quad % half, qy = quad / half;
        const int cx = (mirror & 1) ? regions_per_side - 1 - qx : qx, cy = (mirror & 2) ? regions_per_side - 1 - qy : qy;
        region = (cy * regions_per_side + cx) * 6 + image % 6;
    }
    const int rf = region % 6, cell = region / 6;
    const int rx = cell % regions_per_side, ry = cell / regions_per_side;
    const uint64_t sc_later = (rf >> 1) >= 1 ? ~0ull : 0ull;  // sc is x on the y and z faces, z on the x faces
    const uint64_t tc_later = (rf >> 1) == 2 ? ~0ull : 0ull;  // tc is y on the z faces, z or y (looked at earlier) on the others
    const uint64_t positive = (rf & 1) == 0 ? ~0ull : 0ull;
    const float x_lo = rx == 0 ? -INFINITY : (float)(rx << IRR_REGION_SHIFT), x_hi = rx == regions_per_side - 1 ? INFINITY : (float)((rx + 1) << IRR_REGION_SHIFT);
    const float y_lo = ry == 0 ? -INFINITY : (float)(ry << IRR_REGION_SHIFT), y_hi = ry == regions_per_side - 1 ? INFINITY : (float)((ry + 1) << IRR_REGION_SHIFT);
    for (int t = tid; t < nphi; t += 256) phi_cs[t] = g_irr_phi_cs[t];
    if (tid < ntheta) theta_sc[tid] = g_irr_theta_sc[tid];
    if (tid >= 64 && tid < 68) {  // the face's four neighbours, from cube_fold at t = 0 and t = 1 (it is affine in t)
        const int e = tid - 64, off = (e & 1) ? env_n : -1;
        int f0 = 0, i0 = 0, j0 = 0, f1 = 0, i1 = 0, j1 = 0;
        cube_fold(env_n, rf, e < 2 ? off : 0, e < 2 ? 0 : off, f0, i0, j0);
        if (env_n >= 2) cube_fold(env_n, rf, e < 2 ? off : 1, e < 2 ? 1 : off, f1, i1, j1);
        else { i1 = i0; j1 = j0; }
        edge_map[e][0] = f0; edge_map[e][1] = i0; edge_map[e][2] = i1 - i0; edge_map[e][3] = j0; edge_map[e][4] = j1 - j0;
    }
    if (tid < IRR_GROUP) {
        const int idx = min(group * IRR_GROUP + tid, n_texels - 1);
        const int row = row_begin + idx / n, i = idx % n;
        const int f = row / n, j = row - f * n;
        const f3 N = normalize3(face_dir(f, i, j, n));
        const f3 right = normalize3(mk3(N.z, 0.0f, -N.x));  // normalize(cross((0,1,0), N))
        const f3 up = normalize3(cross3(N, right));
        // kept in the coordinates of the region's face (sc, tc, signed major axis): components swapped and negated,
        // which commutes with the roundings of the sample direction's arithmetic
        const f3 Nf = to_face_axes(rf, N), rightf = to_face_axes(rf, right), upf = to_face_axes(rf, up);
        float *fr = frame[tid];
        fr[0] = Nf.x; fr[1] = Nf.y; fr[2] = Nf.z; fr[3] = rightf.x; fr[4] = rightf.y; fr[5] = rightf.z; fr[6] = upf.x; fr[7] = upf.y; fr[8] = upf.z;
        // the region in face coordinates: base texel i0 = floor(s n - 1/2) in [x0, x1), the first and last regions
        // also take what the clamp sends them
        const float rn = 1.0f / (float)env_n;
        const int x0 = rx << IRR_REGION_SHIFT, x1 = min((rx + 1) << IRR_REGION_SHIFT, env_n), y0 = ry << IRR_REGION_SHIFT, y1 = min((ry + 1) << IRR_REGION_SHIFT, env_n);
        const float u0 = x0 == 0 ? -1.0f : (float)(2 * x0 + 1) * rn - 1.0f, u1 = x1 == env_n ? 1.0f : (float)(2 * x1 + 1) * rn - 1.0f;
        const float v0 = y0 == 0 ? -1.0f : (float)(2 * y0 + 1) * rn - 1.0f, v1 = y1 == env_n ? 1.0f : (float)(2 * y1 + 1) * rn - 1.0f;
        // The region as seen from N, in the coordinates of its face (where N, right, up are Nf, rightf, upf and the region is
        // the pyramid over [u0, u1] x [v0, v1] at height 1; its edges are arcs of great circles).  Polar angle: the
        // farthest point is a corner, the nearest lies on an edge or is N itself.  Azimuth: if N lies outside, the region
        // spans less than half a turn, between the azimuths of two of its corners.  Each bound is widened by IRR_MARGIN,
        // far above the rounding in here and in the running sums that give phi and theta (below 1e-4).
        const float us[4] = {u0, u1, u0, u1}, vs[4] = {v0, v0, v1, v1};
        f3 corner[4];
        float th_max = 0.0f, th_corner[4];
        const f3 centre = mk3(0.5f * (u0 + u1), 0.5f * (v0 + v1), 1.0f);
        const float phc = atan2f(dot3(centre, upf), dot3(centre, rightf));
        float d_lo = 0.0f, d_hi = 0.0f;  // azimuth relative to the centre's
#pragma unroll
        for (int k = 0; k < 4; k++) {
            corner[k] = normalize3(mk3(us[k], vs[k], 1.0f));
            th_corner[k] = acosf(fminf(fmaxf(dot3(corner[k], Nf), -1.0f), 1.0f));
            th_max = fmaxf(th_max, th_corner[k]);
            float d = atan2f(dot3(corner[k], upf), dot3(corner[k], rightf)) - phc;
            d = d > PI_F ? d - 2.0f * PI_F : (d <= -PI_F ? d + 2.0f * PI_F : d);
            d_lo = fminf(d_lo, d); d_hi = fmaxf(d_hi, d);
        }
        const float slack = 4.0f * rn;  // two texels
        const bool n_inside = Nf.z > 0.0f && Nf.x >= (u0 - slack) * Nf.z && Nf.x <= (u1 + slack) * Nf.z && Nf.y >= (v0 - slack) * Nf.z && Nf.y <= (v1 + slack) * Nf.z;
        float th_min = 0.0f;
        if (!n_inside) {
            th_min = 4.0f;
            const int ea[4] = {0, 1, 3, 2}, eb[4] = {1, 3, 2, 0};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const f3 A = corner[ea[e]], B = corner[eb[e]];
                const f3 nrm = normalize3(cross3(A, B));
                // the foot of the perpendicular from N lies on the arc when N is on the inner side of both end planes
                const bool on_arc = dot3(cross3(A, Nf), nrm) >= 0.0f && dot3(cross3(Nf, B), nrm) >= 0.0f;
                const float d = on_arc ? asinf(fminf(__builtin_fabsf(dot3(Nf, nrm)), 1.0f)) : fminf(th_corner[ea[e]], th_corner[eb[e]]);
                th_min = fminf(th_min, d);
            }
        }
        irr_box bx; bx.a0 = 0; bx.na = 0; bx.b0 = 0; bx.nb = 0;
        // N = (0, +-1, 0) has no tangent frame (right = normalize(0) is NaN, as in the shader): every sample direction
        // is then NaN, which the lookup takes to texel (0, 0) of face 1 -- region 1 -- with NaN weights
        const float frame_sum = ((right.x + right.y) + right.z) + ((up.x + up.y) + up.z);
        const float inv = 1.0f / IRR_STEP;
        if (group * IRR_GROUP + tid < n_texels && !(frame_sum == frame_sum)) {
            if (region == 1) { bx.na = nphi; bx.nb = ntheta; bx.a0 = -1; }  // a0 = -1 marks it
        } else if (group * IRR_GROUP + tid < n_texels && (th_min - IRR_MARGIN) * inv < (float)ntheta) {
            bx.b0 = max(0, (int)ceilf((th_min - IRR_MARGIN) * inv));
            bx.nb = min(ntheta - 1, (int)floorf((th_max + IRR_MARGIN) * inv)) - bx.b0 + 1;
            // IRR_MARGIN of arc is this much azimuth at the polar angle where the region comes nearest to N
            const float margin = IRR_MARGIN / fmaxf(sinf(th_min), 1e-6f);
            bool full = n_inside || (d_hi - d_lo) + 2.0f * margin >= 2.0f * PI_F - 2.0f * IRR_STEP;
            if (!full) {
                const float centre_phi = phc < 0.0f ? phc + 2.0f * PI_F : phc;
                const float lo = centre_phi + d_lo - margin, hi = centre_phi + d_hi + margin;
                // phi index 0 also stands for the angle 2 pi, index nphi - 1 for just below it
                const int alo = (int)ceilf(lo * inv), ahi = (int)floorf(hi * inv);
                if (lo < 0.0f && hi >= 2.0f * PI_F) full = true;
                else if (lo < 0.0f) {  // phi in [lo + 2 pi, 2 pi) and in [0, hi]
                    const int w0 = min((int)ceilf((lo + 2.0f * PI_F) * inv), nphi);
                    if (w0 <= ahi + 1) full = true;
                    else { bx.a0 = w0; bx.na = (nphi - w0) + (ahi + 1); }
                } else if (hi >= 2.0f * PI_F) {  // phi in [lo, 2 pi) and in [0, hi - 2 pi]
                    const int e = (int)floorf((hi - 2.0f * PI_F) * inv), first = min(alo, nphi);
                    if (e >= first - 1) full = true;
                    else { bx.a0 = first; bx.na = (nphi - first) + (e + 1); }
                } else { bx.a0 = alo; bx.na = max(min(ahi, nphi - 1) - alo + 1, 0); }
            }
            if (full || bx.na >= nphi) { bx.a0 = 0; bx.na = nphi; }
            if (bx.nb <= 0) bx.na = 0;
        }
        box[tid] = bx;
    }
    __syncthreads();
    // Samples that land in the region are queued per wave (footprint and theta index) and fetched 64 at a time, so
    // that the lanes of a fetch are in use whatever share of a box's cells hit; queue entry k of a pair is fetched
    // and accumulated by lane k mod 64.
    for (int t = w; t < IRR_GROUP; t += 4) {
        const int na = __builtin_amdgcn_readfirstlane(box[t].na);
        float sr = 0.0f, sg = 0.0f, sb = 0.0f;
        int found = 0;
        if (na != 0) {
            int a0 = __builtin_amdgcn_readfirstlane(box[t].a0);
            const int b0 = __builtin_amdgcn_readfirstlane(box[t].b0), nb = __builtin_amdgcn_readfirstlane(box[t].nb);
            const uint64_t every = a0 < 0 ? ~0ull : 0ull;  // a texel without a frame: all of its samples land here
            a0 = max(a0, 0);
            const float *fr = frame[t];
            const f3 N = mk3(fr[0], fr[1], fr[2]), right = mk3(fr[3], fr[4], fr[5]), up = mk3(fr[6], fr[7], fr[8]);
            const float inv_nb = 1.0f / (float)nb;
            const int cells = na * nb;
            // a box that runs over the end of the phi range is walked from phi index 0: the part that wrapped comes first
            const int wrapped = max(a0 + na - nphi, 0);
            int q_head = 0, q_count = 0;
            auto drain = [&](int count) {
                if (lane < count) {
                    const int e = (q_head + lane) & (IRR_QUEUE - 1);
                    const uint4 qe = queue[w][e];
                    const float2 tsc = theta_sc[qe.w];
                    const f3 c = fetch_footprint_on_face<TILED>(env, env_n, rf, (int)(qe.x & 0xFFFFu) - 1, (int)(qe.x >> 16) - 1, __uint_as_float(qe.y), __uint_as_float(qe.z), edge_map);
                    sr = fma_(c.x * tsc.y, tsc.x, sr);
                    sg = fma_(c.y * tsc.y, tsc.x, sg);
                    sb = fma_(c.z * tsc.y, tsc.x, sb);
                }
            };
            // cell kk = base + lane is row q = kk / nb of the box, column kk - q nb; both advance by 64 cells a step without
            // a division.  lane / nb: (lane + 1/2) / nb is at least 1/(2 nb) >= 0.0079 from an integer, the product's error
            // is below 0.002
            const int step_q = (int)(64.5f * inv_nb), step_c = 64 - step_q * nb;
            int q = (int)(((float)lane + 0.5f) * inv_nb), col = lane - q * nb;
            for (int base = 0; base < cells; base += 64) {
                const bool live = q < na;  // rows past the box: the cell is evaluated like the last row's and not counted
                const int qq = min(q, na - 1);
                const int b = b0 + col;
                const int a = qq < wrapped ? qq : a0 + (qq - wrapped);
                q += step_q; col += step_c;
                if (col >= nb) { col -= nb; q++; }
                const float2 pcs = phi_cs[a], tsc = theta_sc[b];
                const float st = tsc.x, ct = tsc.y;
                const float tx = st * pcs.x, ty = st * pcs.y, tz = ct;
                // the direction in the face's coordinates; it lies on the face when cube_face would pick it (ties: z over y over x)
                const float sc = fma_(tz, N.x, fma_(ty, up.x, tx * right.x));
                const float tc = fma_(tz, N.y, fma_(ty, up.y, tx * right.y));
                const float mj = fma_(tz, N.z, fma_(ty, up.z, tx * right.z));
                const float asc = __builtin_fabsf(sc), atc = __builtin_fabsf(tc), ma = __builtin_fabsf(mj);
                // the major axis wins a tie against an axis that cube_face looks at later (z, then y, then x) and loses it
                // against one it looks at earlier; all of it as wave masks, picked by what is uniform for the workgroup
                const uint64_t over_s = (__ballot(ma >= asc) & sc_later) | (__ballot(!(asc >= ma)) & ~sc_later);
                const uint64_t over_t = (__ballot(ma >= atc) & tc_later) | (__ballot(!(atc >= ma)) & ~tc_later);
                const uint64_t side = (__ballot(mj >= 0.0f) & positive) | (__ballot(!(mj <= 0.0f)) & ~positive);
                const float ima = rcp_rn_checked(ma);
                const float s = fma_(sc * ima, 0.5f, 0.5f), tt = fma_(tc * ima, 0.5f, 0.5f);
                const float x = s * (float)env_n - 0.5f, y = tt * (float)env_n - 0.5f;
                // floor(x) clamped to the face lies in the region's columns [x0, x1) when x0 <= x < x1, the first and the
                // last region being open towards the clamp
                const uint64_t inside = __ballot(x >= x_lo) & __ballot(x < x_hi) & __ballot(y >= y_lo) & __ballot(y < y_hi);
                const uint64_t m = __ballot(live) & ((over_s & over_t & side & inside) | every);
                const bool hit = (m >> lane) & 1ull;
                if (hit) {
                    const int e = (q_head + q_count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (IRR_QUEUE - 1);
                    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
                    queue[w][e] = make_uint4((uint32_t)((int)fx + 1) | ((uint32_t)((int)fy + 1) << 16), __float_as_uint(x - fx), __float_as_uint(y - fy), (uint32_t)b);
                }
                const int hits = __popcll(m);
                q_count += hits; found += hits;
                if (q_count >= 64) {
                    drain(64);
                    q_head = (q_head + 64) & (IRR_QUEUE - 1);
                    q_count -= 64;
                }
            }
            drain(q_count);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sr += __shfl_xor(sr, d, 64); sg += __shfl_xor(sg, d, 64); sb += __shfl_xor(sb, d, 64);
            }
        }
        if (lane == 0) { sums[0][t] = sr; sums[1][t] = sg; sums[2][t] = sb; sums[3][t] = (float)found; }
    }
    __syncthreads();
    const size_t padded = (size_t)n_groups * IRR_GROUP;
    partial[((size_t)region * 4 + w) * padded + (size_t)group * IRR_GROUP + lane] = sums[w][lane];
}

// Adds the region sums of every output texel in region order; *uncovered counts the texels whose samples were not
// all found by irradiance_region_kernel (none, unless its cone test has a flaw).
__global__ __launch_bounds__(256) void irradiance_regions_sum_kernel(const float *partial, int n_regions, int n_groups, uint2 *out, int n, int nphi, int ntheta,
                                                                     int row_begin, int n_texels, uint32_t *uncovered)
{
    const int texel = blockIdx.x * 256 + threadIdx.x;
    if (texel >= n_texels) return;
    const size_t padded = (size_t)n_groups * IRR_GROUP;
    float r = 0.0f, g = 0.0f, b = 0.0f, found = 0.0f;
    for (int k = 0; k < n_regions; k++) {
        const float *p = partial + (size_t)k * 4 * padded + texel;
        r += p[0]; g += p[padded]; b += p[2 * padded]; found += p[3 * padded];
    }
    const int count = nphi * ntheta;
    if (found != (float)count) {  // a caller that never looks at the count still cannot take the texel for a result
        atomicAdd(uncovered, 1u);
        r = g = b = __builtin_nanf("");
    }
    const float inv_count = rcp_rn((float)count);
    out[(size_t)row_begin * n + texel] = pack4((PI_F * r) * inv_count, (PI_F * g) * inv_count, (PI_F * b) * inv_count, 1.0f);
}

// ---- irradiance.frag:10-39, summed by region of the environment and by phi row -------------------------------
//
// The arrangement of irradiance_region_kernel with two changes that come from where its time went: finding the samples
// of a (texel, region) pair cost more vector instructions than fetching them, and the fetches sat on the rate at which
// an XCD's L2 hands out cache lines (1.5 lines per sample from 4x4-texel tiles, four load instructions per sample).
//
// (1) The environment is read from "footprint tiles": 4x4-texel tiles of 128 bytes laid at a stride of THREE texels in
// both directions over each face extended by a one-texel apron (the neighbouring faces' texels, as seamless filtering
// folds them).  The 2x2 footprint of a lookup with base texel (i0, j0) then lies in tile ((i0+1)/3, (j0+1)/3), whole:
// one cache line and two 16-byte loads per sample, no edge case except the eight corners of the cube.
//
// (2) A lane takes a phi ROW of a pair, not a cell of its (phi, theta) box: the theta indices of the row that can land
// in the region follow from the four planes that bound it (irr_row_interval), 1.05 cells per sample found against 1.8
// for the box, and the lane walks them with the sample's own arithmetic, adding what lands in the region to ITS sum --
// no queue, no compaction, and the summation order needs none: within a pair every phi row is summed on its own
// (theta ascending), the row sums are added in phi order, the pair sums in region order (order = 3 of the oracle's
// orc_irradiance).  The rows of a workgroup's pairs are sorted by interval length first, so that the lanes of a wave
// finish together.
static constexpr int IRR_ROWS_LDS = 1024;  // row sums a workgroup holds at a time (a texel's rows, at most IRR_MAX_PHI, always fit)
static constexpr int IRR_LEN_BINS = 64;    // interval lengths 0 .. 63 (at most IRR_MAX_THETA - 1)

// Bytes of an n^2-face cube in footprint tiles, and the tile arithmetic (shared by the re-layout kernel and the lookups)
VKR_DEV unsigned div3(unsigned v) { return (v * 43691u) >> 17; }  // exact below 98 304
static inline int64_t footprint_tiles_per_side(int n) { return n / 3 + 1; }

// The cube in footprint tiles.  One lane writes one row of a tile (four texels, 32 bytes); tile (k, m) of a face holds
// the texels (3k - 1 .. 3k + 2) x (3m - 1 .. 3m + 2) of the face, those beyond an edge taken from the neighbouring
// face, those beyond a corner of the cube or beyond the apron zero (never read: see fetch_footprint_tile).
__global__ __launch_bounds__(256) void footprint_tiles_kernel(const uint2 *src, uint4 *dst, int n, unsigned tiles_per_side, unsigned total_rows)
{
    const unsigned g = blockIdx.x * 256u + threadIdx.x;
    if (g >= total_rows) return;
    const unsigned t = g >> 2, row = g & 3u;
    const unsigned per_face = tiles_per_side * tiles_per_side;
    const unsigned f = t / per_face, tf = t - f * per_face;
    const unsigned m = tf / tiles_per_side, k = tf - m * tiles_per_side;
    const int j = (int)(3u * m + row) - 1, i0 = (int)(3u * k) - 1;
    uint2 tx[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int i = i0 + c;
        tx[c] = make_uint2(0u, 0u);
        if (i <= n && j <= n) {
            int of, oi, oj;
            if (cube_fold(n, (int)f, i, j, of, oi, oj)) tx[c] = src[((size_t)of * n + oj) * n + oi];
        }
    }
    dst[(size_t)g * 2u] = make_uint4(tx[0].x, tx[0].y, tx[1].x, tx[1].y);
    dst[(size_t)g * 2u + 1u] = make_uint4(tx[2].x, tx[2].y, tx[3].x, tx[3].y);
}

typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));

// fma(w, t, acc) with t the low / high binary16 half of `packed`, converted on the way in (exact): one instruction
// where a conversion and an fma would be two
VKR_DEV float fma_mix_lo(float w, uint32_t packed, float acc)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(acc));
    return r;
}
VKR_DEV float fma_mix_hi(float w, uint32_t packed, float acc)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed), "v"(acc));
    return r;
}

// A LINEAR lookup in a face stored in footprint tiles, in two steps: where the footprint lies, and the filtering of its
// texels.  (Issuing the loads of a lane's next sample before the current one is filtered was measured and is slower:
// the kernel is bound by the rate at which the L2 hands out cache lines, not by the latency of one.)
// irr_footprint_at: from the unrounded texel coordinates (x, y) = (s n - 1/2, t n - 1/2), the arithmetic of
// texcube_fetch_footprint: the weights' fractions and the byte offset of the footprint's first texel in the face.
// tile_row_step = 128 * tiles_per_side - 96: with (I0, J0) the footprint's base texel shifted by one and (k, m) =
// (I0 / 3, J0 / 3) its tile, that texel lies at ((m * tiles_per_side + k) << 7) + (((J0 - 3m) << 2) + (I0 - 3k) << 3)
// = m * tile_row_step + 104 k + 32 J0 + 8 I0 bytes.  `missing`: at a corner of the cube the number (0 .. 3) of the
// footprint's texel beyond both edges, otherwise -1; only looked for where the region touches a corner of its face
// (`corners`, uniform for the workgroup).
struct irr_footprint { unsigned at; float al, be; int missing; };
VKR_DEV irr_footprint irr_footprint_at(int n, unsigned tile_row_step, bool corners, float x, float y)
{
    irr_footprint fp;
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    fp.al = x - fx; fp.be = y - fy;
    const unsigned I0 = (unsigned)((int)fx + 1), J0 = (unsigned)((int)fy + 1);  // in [0, n]
    const unsigned k = __umul24(I0, 43691u) >> 17, m = __umul24(J0, 43691u) >> 17;  // / 3, exact below 98 304
    fp.at = __umul24(m, tile_row_step) + (__umul24(k, 104u) + ((J0 << 5) + (I0 << 3)));
    fp.missing = -1;
    if (corners) {
        const bool x_edge = I0 == 0u || I0 == (unsigned)n, y_edge = J0 == 0u || J0 == (unsigned)n;
        if (x_edge && y_edge) fp.missing = (int)((I0 == 0u ? 0u : 1u) | (J0 == 0u ? 0u : 2u));
    }
    return fp;
}
// the filtered colour from the footprint's two rows of texels (r0: texels 0 and 1, r1: texels 2 and 3)
VKR_DEV f3 irr_footprint_filter(const irr_footprint &fp, bool corners, u32x4_a8 r0, u32x4_a8 r1)
{
    const float al = fp.al, be = fp.be;
    const float w0 = (1.0f - al) * (1.0f - be), w1 = al * (1.0f - be), w2 = (1.0f - al) * be, w3 = al * be;
    if (__builtin_expect(corners && fp.missing >= 0, 0)) {  // the texel beyond both edges is the mean of the other three
        rgba tx[4] = {unpack4(make_uint2(r0.x, r0.y)), unpack4(make_uint2(r0.z, r0.w)), unpack4(make_uint2(r1.x, r1.y)), unpack4(make_uint2(r1.z, r1.w))};
        float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) if (q != fp.missing) { sr += tx[q].r; sg += tx[q].g; sb += tx[q].b; }
#pragma unroll
        for (int q = 0; q < 4; q++) if (q == fp.missing) { tx[q].r = sr * (1.0f / 3.0f); tx[q].g = sg * (1.0f / 3.0f); tx[q].b = sb * (1.0f / 3.0f); }
        return mk3(fma_(w3, tx[3].r, fma_(w2, tx[2].r, fma_(w1, tx[1].r, w0 * tx[0].r))),
                   fma_(w3, tx[3].g, fma_(w2, tx[2].g, fma_(w1, tx[1].g, w0 * tx[0].g))),
                   fma_(w3, tx[3].b, fma_(w2, tx[2].b, fma_(w1, tx[1].b, w0 * tx[0].b))));
    }
    // the binary16 texels go into the fmas as they are (v_fma_mix_f32 converts on the way in: the same values, twelve
    // conversions fewer); w0 * t0 as fma(w0, t0, -0), which has the product's bits whatever its sign
    const float z = -0.0f;
    return mk3(fma_mix_lo(w3, r1.z, fma_mix_lo(w2, r1.x, fma_mix_lo(w1, r0.z, fma_mix_lo(w0, r0.x, z)))),
               fma_mix_hi(w3, r1.z, fma_mix_hi(w2, r1.x, fma_mix_hi(w1, r0.z, fma_mix_hi(w0, r0.x, z)))),
               fma_mix_lo(w3, r1.w, fma_mix_lo(w2, r1.y, fma_mix_lo(w1, r0.w, fma_mix_lo(w0, r0.y, z)))));
}

// LINEAR, seamless lookup along r in a cube stored in footprint tiles (`tiles`: the whole cube): texcube_rgb_linear
// with the four texels from one tile.
VKR_DEV f3 texcube_rgb_linear_footprint_tiles(const char *tiles, int n, unsigned tiles_per_side, f3 r)
{
    int face; float s, t;
    cube_st(r, face, s, t);
    const float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    // a non-finite direction (a texel without a tangent frame) has NaN coordinates, which the conversion takes to 0
    const irr_footprint fp = irr_footprint_at(n, 128u * tiles_per_side - 96u, true, __builtin_fminf(__builtin_fmaxf(x, -1.0f), (float)n), __builtin_fminf(__builtin_fmaxf(y, -1.0f), (float)n));
    const char *fb = tiles + (size_t)face * tiles_per_side * tiles_per_side * 128u;
    const u32x4_a8 r0 = *(const u32x4_a8 *)(fb + fp.at), r1 = *(const u32x4_a8 *)(fb + fp.at + 32u);
    return irr_footprint_filter(fp, true, r0, r1);
}

// One phi row of a (texel, region) pair: the cells bl .. bh of the row are decided exactly and what lands in the region
// is added to the row's sum (theta ascending).  The face's class is a template argument so that the rules for ties
// between axes are one comparison each: SC_LATER / TC_LATER -- cube_face looks at the axis of sc / tc after the major
// axis (then the major axis wins a tie, otherwise it loses it); POSITIVE -- the face on the positive half axis.
struct irr_row_walk {
    const char *face; int env_n; unsigned tile_row_step; bool corners; float nf;
    float x_lo, x_hi, y_lo, y_hi;
};
struct irr_cell { bool take; irr_footprint fp; float st, ct; };
template <bool SC_LATER, bool TC_LATER, bool POSITIVE>
VKR_DEV irr_cell irr_locate_cell(const irr_row_walk &w, float2 tsc, f3 N, f3 right, f3 up, float2 pcs, bool every)
{
    irr_cell c;
    const float st = tsc.x, ct = tsc.y;
    const float tx = st * pcs.x, ty = st * pcs.y, tz = ct;
    // the direction in the face's coordinates; it lies on the face when cube_face would pick it (ties: z over y over x)
    const float sc = fma_(tz, N.x, fma_(ty, up.x, tx * right.x));
    const float tc = fma_(tz, N.y, fma_(ty, up.y, tx * right.y));
    const float mj = fma_(tz, N.z, fma_(ty, up.z, tx * right.z));
    const float asc = __builtin_fabsf(sc), atc = __builtin_fabsf(tc), ma = __builtin_fabsf(mj);
    const bool over_s = SC_LATER ? ma >= asc : !(asc >= ma);
    const bool over_t = TC_LATER ? ma >= atc : !(atc >= ma);
    const bool side = POSITIVE ? mj >= 0.0f : !(mj <= 0.0f);
    // a direction on this face has |major| >= 1/2 (the direction's length is within rounding of 1), inside the range
    // of the short reciprocal; for any other the quotient is not used
    const float ima = fast_rcp_rn(ma);
    const float s_ = fma_(sc * ima, 0.5f, 0.5f), t_ = fma_(tc * ima, 0.5f, 0.5f);
    const float x = s_ * w.nf - 0.5f, y = t_ * w.nf - 0.5f;
    const bool inside = x >= w.x_lo && x < w.x_hi && y >= w.y_lo && y < w.y_hi;
    c.take = (over_s && over_t && side && inside) || every;
    c.fp = irr_footprint_at(w.env_n, w.tile_row_step, w.corners, x, y);
    c.st = st; c.ct = ct;
    return c;
}
template <bool SC_LATER, bool TC_LATER, bool POSITIVE>
VKR_DEV float4 irr_walk_row(const irr_row_walk &w, const float2 *theta_sc, f3 N, f3 right, f3 up, float2 pcs, int bl, int bh, bool every)
{
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    int found = 0;
    for (int b = bl; b <= bh; b++) {
        const irr_cell c = irr_locate_cell<SC_LATER, TC_LATER, POSITIVE>(w, theta_sc[b], N, right, up, pcs, every);
        if (c.take) {
            const u32x4_a8 r0 = *(const u32x4_a8 *)(w.face + c.fp.at), r1 = *(const u32x4_a8 *)(w.face + c.fp.at + 32u);
            const f3 col = irr_footprint_filter(c.fp, w.corners, r0, r1);
            sr = fma_(col.x * c.ct, c.st, sr);
            sg = fma_(col.y * c.ct, c.st, sg);
            sb = fma_(col.z * c.ct, c.st, sb);
            found++;
        }
    }
    return make_float4(sr, sg, sb, (float)found);
}

// partial: [region][4][texels_padded] as for irradiance_region_kernel; irradiance_regions_sum_kernel adds the regions
__global__ __launch_bounds__(256) void irradiance_rows_kernel(const char *env, int env_n, unsigned tiles_per_side, int n, int nphi, int ntheta, int row_begin, int n_texels,
                                                              float *partial, int regions_per_side, int cells_per_region, int n_regions, int n_groups)
{
    __shared__ float2 phi_cs[IRR_MAX_PHI], theta_sc[IRR_MAX_THETA];
    __shared__ float frame[IRR_GROUP][9];
    __shared__ irr_box box[IRR_GROUP];
    __shared__ int row_base[IRR_GROUP + 1];
    __shared__ float4 row_sum[IRR_ROWS_LDS];      // red, green, blue, samples of a row
    __shared__ uint32_t row_rec[IRR_ROWS_LDS];    // texel | phi index << 6 | first theta << 14 | last theta << 20
    __shared__ uint16_t row_order[IRR_ROWS_LDS];  // rows by falling interval length
    __shared__ uint32_t len_count[IRR_LEN_BINS], len_base[IRR_LEN_BINS];
    __shared__ float4 pair_sum[IRR_GROUP];
    const int tid = threadIdx.x, lane = tid & 63;
    // workgroups go to the XCDs round-robin by id: XCD x works through regions x, x + 8, ... one after the other
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int turn = (kk / n_groups) * 8 + xcd, group = kk % n_groups;
    if (turn >= n_regions) return;
    int region = turn;  // the four mirror images of a region on the six faces weigh the same: every XCD gets three of each
    if ((regions_per_side & 1) == 0) {
        const int half = regions_per_side >> 1, quad = turn / 24, image = turn % 24, mirror = image / 6;
        const int qx = quad % half, qy = quad / half;
        const int cx = (mirror & 1) ? regions_per_side - 1 - qx : qx, cy = (mirror & 2) ? regions_per_side - 1 - qy : qy;
        region = (cy * regions_per_side + cx) * 6 + image % 6;
    }
    const int rf = region % 6, cell = region / 6;
    const int rx = cell % regions_per_side, ry = cell / regions_per_side;
    const irr_region_rect geo = irr_region_geometry(env_n, regions_per_side, cells_per_region, rx, ry);
    // sc is x on the y and z faces (looked at after the major axis), z on the x faces; tc is y on the z faces, z or y
    // (looked at earlier) on the others: the template arguments of irr_walk_row below
    irr_row_walk walk;
    walk.face = env + (size_t)rf * tiles_per_side * tiles_per_side * 128u;
    walk.env_n = env_n; walk.tile_row_step = 128u * tiles_per_side - 96u; walk.nf = (float)env_n;
    walk.corners = (rx == 0 || rx == regions_per_side - 1) && (ry == 0 || ry == regions_per_side - 1);
    walk.x_lo = geo.x_lo; walk.x_hi = geo.x_hi; walk.y_lo = geo.y_lo; walk.y_hi = geo.y_hi;
    for (int t = tid; t < nphi; t += 256) phi_cs[t] = g_irr_phi_cs[t];
    if (tid < ntheta) theta_sc[tid] = g_irr_theta_sc[tid];
    if (tid < IRR_LEN_BINS) len_count[tid] = 0u;
    if (tid < IRR_GROUP) {
        const int idx = min(group * IRR_GROUP + tid, n_texels - 1);
        const int row = row_begin + idx / n, i = idx % n;
        const int f = row / n, j = row - f * n;
        const f3 N = normalize3(face_dir(f, i, j, n));
        const f3 right = normalize3(mk3(N.z, 0.0f, -N.x));  // normalize(cross((0,1,0), N))
        const f3 up = normalize3(cross3(N, right));
        const irr_v3 Nf = irr_to_face_axes(rf, irr_mk(N.x, N.y, N.z)), rightf = irr_to_face_axes(rf, irr_mk(right.x, right.y, right.z)),
                     upf = irr_to_face_axes(rf, irr_mk(up.x, up.y, up.z));
        float *fr = frame[tid];
        fr[0] = Nf.x; fr[1] = Nf.y; fr[2] = Nf.z; fr[3] = rightf.x; fr[4] = rightf.y; fr[5] = rightf.z; fr[6] = upf.x; fr[7] = upf.y; fr[8] = upf.z;
        irr_box bx; bx.a0 = 0; bx.na = 0; bx.b0 = 0; bx.nb = 0;
        // N = (0, +-1, 0) has no tangent frame (right = normalize(0) is NaN, as in the shader): every sample direction
        // is then NaN, which the lookup takes to texel (0, 0) of face 1 -- region 1 -- with NaN weights
        const float frame_sum = ((right.x + right.y) + right.z) + ((up.x + up.y) + up.z);
        if (group * IRR_GROUP + tid < n_texels) {
            if (!(frame_sum == frame_sum)) {
                if (region == 1) { bx.na = nphi; bx.nb = ntheta; bx.a0 = -1; }  // a0 = -1 marks it
            } else bx = irr_pair_box(Nf, rightf, upf, geo.u0, geo.u1, geo.v0, geo.v1, 4.0f / (float)env_n, nphi, ntheta);
        }
        box[tid] = bx;
        int incl = bx.na;  // rows of the texels before this one
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
        row_base[tid + 1] = incl;
        if (tid == 0) row_base[0] = 0;
        pair_sum[tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    for (int t0 = 0; t0 < IRR_GROUP;) {
        // the texels [t0, t1) whose rows fit the row buffer together
        int t1 = IRR_GROUP;
        if (row_base[IRR_GROUP] - row_base[t0] > IRR_ROWS_LDS) {
            int lo = t0 + 1, hi = IRR_GROUP;  // row_base[lo] - row_base[t0] fits, row_base[hi] - .. does not
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_base[mid] - row_base[t0] <= IRR_ROWS_LDS) lo = mid; else hi = mid; }
            t1 = lo;
        }
        const int base = row_base[t0], rows = row_base[t1] - base;
        // every row: its texel, its phi index, the theta interval that can land in the region; counted by interval length
        for (int q = tid; q < rows; q += 256) {
            const int g = base + q;
            int lo = t0, hi = t1;  // row_base[lo] <= g < row_base[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_base[mid] <= g) lo = mid; else hi = mid; }
            const int t = lo, r = g - row_base[t];
            const irr_box bx = box[t];
            const int a0 = max(bx.a0, 0), wrapped = max(a0 + bx.na - nphi, 0);  // a box that runs over the end of the phi range: the part that wrapped comes first
            const int a = r < wrapped ? r : a0 + (r - wrapped);
            int bl = bx.b0, bh = bx.b0 + bx.nb - 1;
            if (bx.a0 >= 0) {
                const float *fr = frame[t];
                const float2 pcs = phi_cs[a];
                int il, ih;
                irr_row_interval(irr_mk(fr[0], fr[1], fr[2]), irr_mk(fr[3], fr[4], fr[5]), irr_mk(fr[6], fr[7], fr[8]), pcs.x, pcs.y, geo.U0, geo.U1, geo.V0, geo.V1, &il, &ih);
                bl = max(bl, il); bh = min(bh, ih);
            }
            const int len = max(bh - bl + 1, 0);
            if (len == 0) { bl = 1; bh = 0; row_sum[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            row_rec[q] = (uint32_t)t | ((uint32_t)a << 6) | ((uint32_t)bl << 14) | ((uint32_t)bh << 20);
            atomicAdd(&len_count[len], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // longest rows first; rows without an interval are left out
            const uint32_t c = tid == 0 ? 0u : len_count[IRR_LEN_BINS - tid];  // lane k: length 64 - k
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
            if (tid > 0) len_base[IRR_LEN_BINS - tid] = incl - c;
            if (tid == 63) len_base[0] = incl;  // = the number of rows with an interval
        }
        __syncthreads();
        for (int q = tid; q < rows; q += 256) {
            const uint32_t rec = row_rec[q];
            const int len = max((int)((rec >> 20) & 63u) - (int)((rec >> 14) & 63u) + 1, 0);
            if (len > 0) row_order[atomicAdd(&len_base[len], 1u)] = (uint16_t)q;
        }
        __syncthreads();
        const int busy = (int)len_base[0];
        // a lane walks its row's interval: every cell decided exactly, what lands in the region added to the row's sum
        for (int s = tid; s < busy; s += 256) {
            const int q = row_order[s];
            const uint32_t rec = row_rec[q];
            const int t = (int)(rec & 63u), a = (int)((rec >> 6) & 255u), bl = (int)((rec >> 14) & 63u), bh = (int)((rec >> 20) & 63u);
            const bool every = box[t].a0 < 0;  // a texel without a frame: all of its samples land here
            const float *fr = frame[t];
            const f3 N = mk3(fr[0], fr[1], fr[2]), right = mk3(fr[3], fr[4], fr[5]), up = mk3(fr[6], fr[7], fr[8]);
            const float2 pcs = phi_cs[a];
            float4 sum;
            switch (rf) {
            case 0: sum = irr_walk_row<false, false, true>(walk, theta_sc, N, right, up, pcs, bl, bh, every); break;
            case 1: sum = irr_walk_row<false, false, false>(walk, theta_sc, N, right, up, pcs, bl, bh, every); break;
            case 2: sum = irr_walk_row<true, false, true>(walk, theta_sc, N, right, up, pcs, bl, bh, every); break;
            case 3: sum = irr_walk_row<true, false, false>(walk, theta_sc, N, right, up, pcs, bl, bh, every); break;
            case 4: sum = irr_walk_row<true, true, true>(walk, theta_sc, N, right, up, pcs, bl, bh, every); break;
            default: sum = irr_walk_row<true, true, false>(walk, theta_sc, N, right, up, pcs, bl, bh, every); break;
            }
            row_sum[q] = sum;
        }
        __syncthreads();
        {   // the rows of a pair in phi order, one lane per texel and channel (the fourth: the number of samples)
            const int t = t0 + (tid >> 2), ch = tid & 3;
            if (t < t1) {
                const float *rs = (const float *)row_sum + ch;
                float acc = 0.0f;
                for (int q = row_base[t] - base, e = row_base[t + 1] - base; q < e; q++) acc += rs[4 * q];
                ((float *)&pair_sum[t])[ch] = acc;
            }
            if (tid < IRR_LEN_BINS) len_count[tid] = 0u;
        }
        __syncthreads();
        t0 = t1;
    }
    const size_t padded = (size_t)n_groups * IRR_GROUP;
    const int w = tid >> 6;
    partial[((size_t)region * 4 + w) * padded + (size_t)group * IRR_GROUP + lane] = ((const float *)&pair_sum[lane])[w];
}

// ---- prefilter.frag / brdf.frag shared pieces -----------------------------------------------

// prefilter.frag:65-73
VKR_DEV float radical_inverse(uint32_t bits)
{
    return (float)__builtin_bitreverse32(bits) * 2.3283064365386963e-10f;
}

// (cos theta, sin theta) of GGX sample s at roughness r (prefilter.frag:82-86); lane-independent
VKR_DEV float2 ggx_theta(int s, float rough)
{
    float a = rough * rough;
    float xi_y = radical_inverse((uint32_t)s);
    float cos_t = sqrt_rn_checked((1.0f - xi_y) * rcp_rn_checked(fma_(a * a - 1.0f, xi_y, 1.0f)));
    float sin_t = sqrt_rn_checked(fma_(-cos_t, cos_t, 1.0f));
    return make_float2(cos_t, sin_t);
}

// prefilter.frag:21-63, one workgroup per two or sixteen neighbouring output texels; with sixteen the 1024 samples go in
// eight passes of 128.  Per pass every lane evaluates its samples (environment colour and N.L) for its texels into
// LDS -- the sample's place on the GGX lobe depends on the roughness only and is computed once per lane; one lane per
// texel and channel -- red, green, blue and the weight total -- then adds the pass to its running sum in sample
// order, which is the shader's single running sum.  Sixteen texels where the launch is large and bound by instruction
// issue (mip 0: the 64 summing lanes cost the instructions, and the other waves the waiting, that eight did: 0.80 ->
// 0.52 ms); two for the smaller mips, which are bound by cache-line traffic and want the occupancy, and so that the
// small ones fill the chip.
// One launch may hold several mips (the small ones together fill the chip where each alone would not): per mip its
// image, face size, roughness, the rows it computes and the number of texel groups each XCD takes of it.
static constexpr int PREFILTER_MAX_MIPS = 8;
struct PrefilterMips {
    uint2 *out[PREFILTER_MAX_MIPS]; int n[PREFILTER_MAX_MIPS]; float rough[PREFILTER_MAX_MIPS];
    int row_begin[PREFILTER_MAX_MIPS], n_texels[PREFILTER_MAX_MIPS], per_xcd[PREFILTER_MAX_MIPS];
    int count;
};
template <int LAYOUT, int PREFILTER_TEXELS>
__global__ __launch_bounds__(256) void prefilter_texel_kernel(const void *env, int env_n, const PrefilterMips mips)
{
    constexpr int PREFILTER_PASS = 2 * SAMPLE_COUNT / PREFILTER_TEXELS;  // 32 KB of terms either way
    // the terms of a pass: [0..2] environment colour, [3] N.L; at roughness 0 the same memory holds the terms of the samples
    // that need evaluating at all, [sample][texel][4] (see below)
    constexpr int ODD_MAX = 144;  // 137 at roughness 0
    constexpr int TERM_FLOATS = PREFILTER_TEXELS * 4 * PREFILTER_PASS > ODD_MAX * PREFILTER_TEXELS * 4 ? PREFILTER_TEXELS * 4 * PREFILTER_PASS : ODD_MAX * PREFILTER_TEXELS * 4;
    __shared__ __align__(16) float term_store[TERM_FLOATS];
    float (*const term)[4][PREFILTER_PASS] = (float (*)[4][PREFILTER_PASS])term_store;
    __shared__ __align__(16) float ones[16];
    __shared__ float result[PREFILTER_TEXELS][4];
    const int tid = threadIdx.x;
    const unsigned tiles_per_side = (unsigned)env_n / 3u + 1u;  // of the footprint-tile layout
    // Workgroups go to the 8 XCDs round-robin by id.  Of every mip XCD x takes the groups [x * per_xcd, (x + 1) * per_xcd)
    // in row-major order, so the workgroups resident on an XCD at any time are neighbours on the cube, their lobes
    // overlap, and the XCD's L2 serves most of their lookups; an XCD works through its share of the first mip of the
    // launch, then of the next (the largest first).
    int mip = 0, k = (int)(blockIdx.x >> 3);
    while (mip + 1 < mips.count && k >= mips.per_xcd[mip]) { k -= mips.per_xcd[mip]; mip++; }
    uint2 *const out = mips.out[mip];
    const int n = mips.n[mip], row_begin = mips.row_begin[mip], n_texels = mips.n_texels[mip];
    const float rough = mips.rough[mip];
    const int first = PREFILTER_TEXELS * ((int)(blockIdx.x & 7u) * mips.per_xcd[mip] + k);
    if (first >= n_texels) return;
    if (tid < 16) ones[tid] = 1.0f;
    __shared__ float frame[PREFILTER_TEXELS][12];  // N, tangent, bitangent of each texel
    if (tid < PREFILTER_TEXELS) {
        const int texel = min(first + tid, n_texels - 1);  // a group beyond the last texel repeats it and stores it once
        const int i = texel % n, row = row_begin + texel / n;
        const int f = row / n, j = row - f * n;
        const f3 N = normalize3(face_dir(f, i, j, n));
        const f3 tangent = __builtin_fabsf(N.z) < 0.999f ? normalize3(mk3(-N.y, N.x, 0.0f))   // cross((0,0,1), N)
                                                         : normalize3(mk3(0.0f, -N.z, N.y));  // cross((1,0,0), N)
        const f3 bitangent = cross3(N, tangent);
        float *fr = frame[tid];
        fr[0] = N.x; fr[1] = N.y; fr[2] = N.z; fr[3] = tangent.x; fr[4] = tangent.y; fr[5] = tangent.z; fr[6] = bitangent.x; fr[7] = bitangent.y; fr[8] = bitangent.z;
    }
    __syncthreads();
    // the sixteen summing lanes: texel tid / 4, channel tid % 4 (3: totalWeight += NdotL, as a product with a block of
    // ones that the lane keeps re-reading, so that all sixteen run the same multiply-adds)
    const int sum_lane = tid;
    const bool sums = tid < 4 * PREFILTER_TEXELS;  // the first wave, or part of it
    const int sum_t = (sum_lane >> 2) & (PREFILTER_TEXELS - 1), sum_ch = sum_lane & 3;
    const float4 *const sum_mine = sum_ch == 3 ? (const float4 *)ones : (const float4 *)term[sum_t][sum_ch];
    const float4 *const sum_wts = (const float4 *)term[sum_t][3];
    const int sum_stride = sum_ch == 3 ? 0 : 1;
    // Roughness 0 (mip 0): the lobe has collapsed onto N.  In binary32 the shader's cos theta = sqrt((1 - y) * (1 / (1 - y)))
    // comes out as exactly 1 (and sin theta as 0) for 887 of the 1024 samples -- for the other 137 the product rounds to
    // 1 - 2^-24 and sin theta is 3.45e-4, a different half vector per sample.  A sample with (cos, sin) = (1, 0) has the
    // half vector N * 1 + (+-0) + (+-0) = N whenever no component of N is zero (a sum with a zero operand is exact), so
    // all of them share one L, one N.L and one environment colour per texel.  These are computed once, the other samples
    // are evaluated in one sweep with all their lookups in flight, and the running sums then take their 1024 terms in
    // sample order from the two sources -- the same terms in the same order as the general path below.
    __shared__ float same[PREFILTER_TEXELS][4];
    __shared__ float4 odd_h[ODD_MAX];                    // (hx, hy, hz, sample index) of the samples that need evaluating, in sample order
    __shared__ unsigned long long odd_mask[SAMPLE_COUNT / 64];  // bit s & 63 of word s >> 6: sample s needs evaluating
    __shared__ int shared_ok;
    auto evaluate = [&](int t, float hx, float hy, float hz, float &r, float &g, float &b, float &w) {  // one sample of texel t
        const float *fr = frame[t];
        const f3 N = mk3(fr[0], fr[1], fr[2]), tangent = mk3(fr[3], fr[4], fr[5]), bitangent = mk3(fr[6], fr[7], fr[8]);
        const f3 H = normalize3_fast(mk3(fma_(N.x, hz, fma_(bitangent.x, hy, tangent.x * hx)),
                                         fma_(N.y, hz, fma_(bitangent.y, hy, tangent.y * hx)),
                                         fma_(N.z, hz, fma_(bitangent.z, hy, tangent.z * hx))));
        const float vh2 = 2.0f * dot3(N, H);  // V = N
        const f3 L = normalize3_fast(mk3(fma_(vh2, H.x, -N.x), fma_(vh2, H.y, -N.y), fma_(vh2, H.z, -N.z)));
        const float NdotL = max0(dot3(N, L));
        f3 c = mk3(0.0f, 0.0f, 0.0f);
        if (NdotL > 0.0f)  // the shader's mip level is clamped to 0 by the sampler
            c = LAYOUT == VKR_ENV_LAYOUT_FOOTPRINT_TILES ? texcube_rgb_linear_footprint_tiles((const char *)env, env_n, tiles_per_side, L)
                                                         : texcube_rgb_linear<LAYOUT == VKR_ENV_LAYOUT_TILED_4X4>(env, env_n, L);
        r = c.x; g = c.y; b = c.z; w = NdotL;
    };
    float acc = 0.0f;
    bool summed = false;
    // (Sixteen texels per workgroup only: with two, the one pass of the general path already spreads the samples over all
    // lanes and its sums stream from LDS; the sweep-then-sum form measured 0.3 ms slower there.)
    if (PREFILTER_TEXELS == 16 && rough == 0.0f) {  // the same for the whole workgroup
        static_assert(SAMPLE_COUNT == 1024, "four samples per lane below");
        if (tid == 0) shared_ok = 1;
        __syncthreads();
        bool odd[4];
        float2 tcs4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {  // sample m * 256 + tid: a wave holds 64 consecutive samples, one word of the mask
            tcs4[m] = ggx_theta(m * 256 + tid, rough);
            odd[m] = !(tcs4[m].x == 1.0f && tcs4[m].y == 0.0f);
            const unsigned long long word = __builtin_amdgcn_ballot_w64(odd[m]);
            if ((tid & 63) == 0) odd_mask[m * 4 + (tid >> 6)] = word;
        }
        if (tid < PREFILTER_TEXELS) {
            const float *fr = frame[tid];
            if (fr[0] == 0.0f || fr[1] == 0.0f || fr[2] == 0.0f) shared_ok = 0;  // a zero component: the signs of the zeros could matter
            evaluate(tid, 0.0f, 0.0f, 1.0f, same[tid][0], same[tid][1], same[tid][2], same[tid][3]);
        }
        __syncthreads();
        int n_odd = 0;
        for (int k = 0; k < SAMPLE_COUNT / 64; k++) n_odd += __builtin_popcountll(odd_mask[k]);
        if (shared_ok != 0 && n_odd <= ODD_MAX) {
#pragma unroll
            for (int m = 0; m < 4; m++)
                if (odd[m]) {
                    const int s = m * 256 + tid, word = s >> 6;
                    int rank = __builtin_popcountll(odd_mask[word] & ((1ull << (s & 63)) - 1ull));
                    for (int k = 0; k < word; k++) rank += __builtin_popcountll(odd_mask[k]);
                    const float2 pcs = g_hammersley_cs[s];
                    odd_h[rank] = make_float4(pcs.x * tcs4[m].y, pcs.y * tcs4[m].y, tcs4[m].x, __int_as_float(s));
                }
            __syncthreads();
            float (*const odd_term)[PREFILTER_TEXELS][4] = (float (*)[PREFILTER_TEXELS][4])term_store;
            // three items per trip, so that their environment lookups are in flight together
#pragma unroll 1
            for (int item = tid; item < n_odd * PREFILTER_TEXELS; item += 3 * 256) {
                float r[3], g[3], b[3], w[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int it = min(item + u * 256, n_odd * PREFILTER_TEXELS - 1);  // past the end: the last item again, stored once
                    const float4 h = odd_h[it / PREFILTER_TEXELS];
                    evaluate(it % PREFILTER_TEXELS, h.x, h.y, h.z, r[u], g[u], b[u], w[u]);
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int it = item + u * 256;
                    if (it < n_odd * PREFILTER_TEXELS) {
                        float *o = odd_term[it / PREFILTER_TEXELS][it % PREFILTER_TEXELS];
                        o[0] = r[u]; o[1] = g[u]; o[2] = b[u]; o[3] = w[u];
                    }
                }
            }
            __syncthreads();
            if (sums) {
                // One dependent fma per sample, as below; channel 3 adds N.L * 1.  Between two samples that were evaluated lies
                // a run of shared terms: the same fma over and over, with nothing to load and a scalar trip count.
                const float c_same = sum_ch == 3 ? 1.0f : same[sum_t][sum_ch], w_same = same[sum_t][3];
                int done = 0;  // samples added so far
                float c_next = 0.0f, w_next = 0.0f;
                if (n_odd > 0) { c_next = sum_ch == 3 ? 1.0f : odd_term[0][sum_t][sum_ch]; w_next = odd_term[0][sum_t][3]; }
                int upto_next = n_odd > 0 ? __float_as_int(odd_h[0].w) : SAMPLE_COUNT;  // index of the next evaluated sample
                for (int k = 0; k <= n_odd; k++) {
                    const int upto = __builtin_amdgcn_readfirstlane(upto_next);
                    const float c = c_next, w = w_next;
                    // the terms and the index after these are fetched while this run is added
                    if (k + 1 < n_odd) { c_next = sum_ch == 3 ? 1.0f : odd_term[k + 1][sum_t][sum_ch]; w_next = odd_term[k + 1][sum_t][3]; upto_next = __float_as_int(odd_h[k + 1].w); }
                    else upto_next = SAMPLE_COUNT;
                    int run = __builtin_amdgcn_readfirstlane(upto - done);
                    // straight-line code for the run: a taken branch costs several fmas, and the runs are short (6.5 on average)
#define VKR_SAME acc = fma_(c_same, w_same, acc);
                    for (; run >= 8; run -= 8) { VKR_SAME VKR_SAME VKR_SAME VKR_SAME VKR_SAME VKR_SAME VKR_SAME VKR_SAME }
                    if (run & 4) { VKR_SAME VKR_SAME VKR_SAME VKR_SAME }
                    if (run & 2) { VKR_SAME VKR_SAME }
                    if (run & 1) { VKR_SAME }
#undef VKR_SAME
                    if (k < n_odd) acc = fma_(c, w, acc);
                    done = upto + 1;
                }
            }
            summed = true;
        }
    }
    if (!summed)
    for (int half = 0; half < SAMPLE_COUNT / PREFILTER_PASS; half++) {
        if (half) __syncthreads();  // the sums of the first half have been taken
        // a pass of 256 samples or more: every lane takes all texels of its samples; a shorter one: the lanes share the
        // texels of a sample between them
        constexpr int LANES = PREFILTER_PASS < 256 ? PREFILTER_PASS : 256, SHARE = 256 / LANES, MINE = PREFILTER_TEXELS / SHARE;
        const int t_first = (tid / LANES) * MINE;
#pragma unroll 1
        for (int m = 0; m < PREFILTER_PASS / LANES; m++) {
            const int sl = tid % LANES + LANES * m, s = half * PREFILTER_PASS + sl;
            const float2 pcs = g_hammersley_cs[s];
            const float2 tcs = ggx_theta(s, rough);
            const float hx = pcs.x * tcs.y, hy = pcs.y * tcs.y, hz = tcs.x;
#pragma unroll 1
            for (int t = t_first; t < t_first + MINE; t++) {
                const float *fr = frame[t];
                const f3 N = mk3(fr[0], fr[1], fr[2]), tangent = mk3(fr[3], fr[4], fr[5]), bitangent = mk3(fr[6], fr[7], fr[8]);
                const f3 H = normalize3_fast(mk3(fma_(N.x, hz, fma_(bitangent.x, hy, tangent.x * hx)),
                                                 fma_(N.y, hz, fma_(bitangent.y, hy, tangent.y * hx)),
                                                 fma_(N.z, hz, fma_(bitangent.z, hy, tangent.z * hx))));
                const float vh2 = 2.0f * dot3(N, H);  // V = N
                const f3 L = normalize3_fast(mk3(fma_(vh2, H.x, -N.x), fma_(vh2, H.y, -N.y), fma_(vh2, H.z, -N.z)));
                const float NdotL = max0(dot3(N, L));
                f3 c = mk3(0.0f, 0.0f, 0.0f);
                if (NdotL > 0.0f)  // the shader's mip level is clamped to 0 by the sampler
                    c = LAYOUT == VKR_ENV_LAYOUT_FOOTPRINT_TILES ? texcube_rgb_linear_footprint_tiles((const char *)env, env_n, tiles_per_side, L)
                                                                 : texcube_rgb_linear<LAYOUT == VKR_ENV_LAYOUT_TILED_4X4>(env, env_n, L);
                term[t][0][sl] = c.x; term[t][1][sl] = c.y; term[t][2][sl] = c.z; term[t][3][sl] = NdotL;
            }
        }
        __syncthreads();
        if (sums) {
            // The running sum is one dependent fma per sample; the terms are read sixteen samples ahead, so that a step
            // costs the fma's latency and not an LDS round trip.  A sample the shader skips (N.L <= 0) was stored as
            // colour 0, weight 0: adding 0 * 0 leaves the running sum's bits unchanged, so no test is needed here.
            constexpr int AHEAD = 4, QUADS = PREFILTER_PASS / 4;
            float4 c[AHEAD], wt[AHEAD];
#pragma unroll
            for (int u = 0; u < AHEAD; u++) { c[u] = sum_mine[u]; wt[u] = sum_wts[u]; }
            for (int q = 0; q < QUADS; q += AHEAD) {
                float4 cn[AHEAD], wn[AHEAD];
                const int qn = min(q + AHEAD, QUADS - AHEAD);
                const float4 *next = sum_mine + qn * sum_stride;
#pragma unroll
                for (int u = 0; u < AHEAD; u++) { cn[u] = next[u]; wn[u] = sum_wts[qn + u]; }
#pragma unroll
                for (int u = 0; u < AHEAD; u++) {
                    acc = fma_(c[u].x, wt[u].x, acc);
                    acc = fma_(c[u].y, wt[u].y, acc);
                    acc = fma_(c[u].z, wt[u].z, acc);
                    acc = fma_(c[u].w, wt[u].w, acc);
                }
#pragma unroll
                for (int u = 0; u < AHEAD; u++) { c[u] = cn[u]; wt[u] = wn[u]; }
            }
        }
    }
    if (sums) result[sum_t][sum_ch] = acc;
    __syncthreads();
    if (tid < PREFILTER_TEXELS && first + tid < n_texels) {
        const int texel = first + tid;
        const int i = texel % n, row = row_begin + texel / n;
        const float iw = rcp_rn(result[tid][3]);
        out[(size_t)row * n + i] = pack4(result[tid][0] * iw, result[tid][1] * iw, result[tid][2] * iw, 1.0f);
    }
}

// brdf.frag:81-119, one lane per texel; a workgroup works on one row (one roughness).
// The half vector of sample s depends on the sample and on the roughness only, not on N.V: with N = (0,0,1)
// the shader's tangent frame is (0,-1,0), (1,0,-0) (brdf.frag:52-56), so its sampleVec is (hy, -hx, hz).  The
// workgroup normalises the row's 1024 half vectors once into LDS; the sample loop then reads one broadcast
// 16-byte entry per sample and does only the work that depends on the lane's N.V.
#ifndef VKR_LUT_UNROLL
#define VKR_LUT_UNROLL 4
#endif
__global__ __launch_bounds__(256) void brdf_lut_kernel(uint32_t *out, int n, int row_begin, int row_end)
{
    __shared__ float4 half_vec[SAMPLE_COUNT];  // normalised H and N.H = max(H.z, 0)
    int row = row_begin + blockIdx.y;
    float rn = rcp_rn((float)n);
    float rough = ((float)row + 0.5f) * rn;
    for (int s = threadIdx.x; s < SAMPLE_COUNT; s += 256) {
        float2 pcs = g_hammersley_cs[s];
        float2 tcs = ggx_theta(s, rough);
        float hx = pcs.x * tcs.y, hy = pcs.y * tcs.y, hz = tcs.x;
        f3 H = normalize3_fast(mk3(hy, -hx, hz));
        half_vec[s] = make_float4(H.x, H.y, H.z, max0(H.z));
    }
    __syncthreads();
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || row >= row_end) return;
    float NdotV = ((float)i + 0.5f) * rn;
    float Vx = sqrt_rn(fma_(-NdotV, NdotV, 1.0f)), Vz = NdotV;  // V.y = 0 drops out of dot(V,H) and of L.y
    float k = (rough * rough) * 0.5f;
    float one_minus_k = 1.0f - k;
    float gV = max0(Vz) * rcp_rn(fma_(max0(Vz), one_minus_k, k));  // GeometrySchlickGGX(NdotV), brdf.frag:60-69
    float A = 0.0f, B = 0.0f;
#pragma unroll VKR_LUT_UNROLL
    for (int s = 0; s < SAMPLE_COUNT; s++) {
        const float4 H = half_vec[s];
        float vh = fma_(Vz, H.z, Vx * H.x);
        float vh2 = 2.0f * vh;
        // L = normalize(2 (V.H) H - V): V and H are unit vectors up to rounding, so |L|^2 is within a few ulps of 1
        // and the short square-root / reciprocal sequences are inside their exact domain (same bits as the IEEE ones)
        f3 Lr = mk3(fma_(vh2, H.x, -Vx), vh2 * H.y, fma_(vh2, H.z, -Vz));
        float Lz = Lr.z * fast_rcp_rn(fast_sqrt_rn(dot3(Lr, Lr)));
        float NdotL = max0(Lz), NdotH = H.w, VdotH = max0(vh);
        // denominators: N.L (1-k) + k lies in [k, 1] with k >= 1/(8 n^2); N.H >= 1/32 for every sample (xi <= 1023/1024)
        // and N.V >= 1/(2n): both are normal numbers with normal reciprocals
        float G = (NdotL * fast_rcp_rn(fma_(NdotL, one_minus_k, k))) * gV;
        float G_vis = (G * VdotH) * fast_rcp_rn(NdotH * NdotV);
        float Fc = pow5(1.0f - VdotH);
        // a sample the shader skips (N.L <= 0) adds x * 0 to sums that are never negative: their bits stay as they are
        G_vis = NdotL > 0.0f ? G_vis : 0.0f;
        A = fma_(1.0f - Fc, G_vis, A);
        B = fma_(Fc, G_vis, B);
    }
    half2_t o; o.x = to_f16(A * (1.0f / 1024.0f)); o.y = to_f16(B * (1.0f / 1024.0f));
    out[(size_t)row * n + i] = __builtin_bit_cast(uint32_t, o);
}

// Row-major RGBA16F faces -> 4x4-texel tiles.  One lane moves one 32-byte tile row; eight lanes
// write two consecutive tiles (256 contiguous bytes).
__global__ __launch_bounds__(256) void tile_rgba16f_kernel(const uint4 *src, uint4 *dst, int n, unsigned tiles_per_face, unsigned total_tiles)
{
    const unsigned g = blockIdx.x * 256u + threadIdx.x;
    const unsigned t = g >> 2, row = g & 3u;  // tile over all faces, row inside the tile
    if (t >= total_tiles) return;
    const unsigned face = t / tiles_per_face, tf = t - face * tiles_per_face;
    const unsigned tiles_x = (unsigned)n >> 2;
    const unsigned ty = tf / tiles_x, tx = tf - ty * tiles_x;
    const size_t texel = (size_t)face * n * n + (size_t)(ty * 4u + row) * n + tx * 4u;  // four texels = two uint4
    dst[(size_t)g * 2u] = src[texel >> 1];
    dst[(size_t)g * 2u + 1u] = src[(texel >> 1) + 1u];
}

static int32_t check_env_layout(const char *who, int env_n, int layout)
{
    VKR_REQUIRE(layout == VKR_ENV_LAYOUT_LINEAR || layout == VKR_ENV_LAYOUT_TILED_4X4, "%s: unknown environment layout %d", who, layout);
    VKR_REQUIRE(layout == VKR_ENV_LAYOUT_LINEAR || env_n % 4 == 0, "%s: the tiled environment layout needs a face size that is a multiple of 4 (got %d)", who, env_n);
    return VKR_OK;
}

static int32_t check_rows(const char *who, int n, int total_rows, int row_begin, int row_end)
{
    VKR_REQUIRE(n > 0, "%s: bad size %d", who, n);
    VKR_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= total_rows, "%s: rows [%d,%d) outside [0,%d)", who, row_begin, row_end, total_rows);
    return VKR_OK;
}

}  // namespace vkr

extern "C" int32_t vkr_ibl_equirect_to_cube(const void *src, int32_t src_w, int32_t src_h, int32_t src_format, void *cube, int32_t n,
                                            int32_t row_begin, int32_t row_end, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(src && cube, "vkr_ibl_equirect_to_cube: an image is null");
    VKR_REQUIRE(src_w > 0 && src_h > 0, "vkr_ibl_equirect_to_cube: bad source size %dx%d", src_w, src_h);
    VKR_REQUIRE(src_format == VKR_SRC_RGBA8_SRGB || src_format == VKR_SRC_RGBA16F, "vkr_ibl_equirect_to_cube: unknown source format %d", src_format);
    int32_t rc = check_rows("vkr_ibl_equirect_to_cube", n, 6 * n, row_begin, row_end);
    if (rc != VKR_OK) return rc;
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    if (row_end == row_begin) return VKR_OK;
    dim3 grid((n + 255) / 256, row_end - row_begin);
    hipLaunchKernelGGL(equirect_kernel, grid, dim3(256), 0, as_stream(stream), src, src_w, src_h, src_format, (uint2 *)cube, n, 1.0f / (float)n, row_begin, row_end);
    return launch_status("vkr_ibl_equirect_to_cube");
}

extern "C" int32_t vkr_tile_rgba16f_faces(const void *linear, void *tiled, int32_t n_faces, int32_t n, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(linear && tiled, "vkr_tile_rgba16f_faces: an image is null");
    VKR_REQUIRE(n_faces >= 0 && n > 0 && n % 4 == 0 && n <= 16384, "vkr_tile_rgba16f_faces: %d faces of size %d (size must be a multiple of 4, at most 16384)", n_faces, n);
    const size_t bytes = (size_t)n_faces * n * n * 8;
    const char *a = (const char *)linear, *b = (const char *)tiled;
    VKR_REQUIRE(a + bytes <= b || b + bytes <= a, "vkr_tile_rgba16f_faces: source and destination overlap");
    const uint64_t tiles_per_face = (uint64_t)(n / 4) * (n / 4), total = tiles_per_face * (uint64_t)n_faces;
    VKR_REQUIRE(total < (1ull << 29), "vkr_tile_rgba16f_faces: too many texels for one call");
    if (total == 0) return VKR_OK;
    hipLaunchKernelGGL(tile_rgba16f_kernel, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const uint4 *)linear, (uint4 *)tiled, n, (unsigned)tiles_per_face, (unsigned)total);
    return launch_status("vkr_tile_rgba16f_faces");
}

extern "C" int32_t vkr_ibl_irradiance(const void *env, int32_t env_n, int32_t env_layout, void *out, int32_t n, int32_t order, int32_t row_begin, int32_t row_end, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(env && out && env_n > 0, "vkr_ibl_irradiance: an image is null or env size is bad");
    VKR_REQUIRE(order != VKR_IRRADIANCE_BY_REGION, "vkr_ibl_irradiance: the by-region order needs a workspace, call vkr_ibl_irradiance_by_region");
    VKR_REQUIRE(order == VKR_IRRADIANCE_SEQUENTIAL || order == VKR_IRRADIANCE_PHI_SLICES, "vkr_ibl_irradiance: unknown summation order %d", order);
    int32_t rc = check_rows("vkr_ibl_irradiance", n, 6 * n, row_begin, row_end);
    if (rc != VKR_OK) return rc;
    if ((rc = check_env_layout("vkr_ibl_irradiance", env_n, env_layout)) != VKR_OK) return rc;
    const bool tiled = env_layout == VKR_ENV_LAYOUT_TILED_4X4;
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    if (row_end == row_begin) return VKR_OK;
    if (order == VKR_IRRADIANCE_PHI_SLICES) {
        dim3 grid(n, row_end - row_begin);
        hipLaunchKernelGGL(tiled ? irradiance_slices_kernel<true> : irradiance_slices_kernel<false>, grid, dim3(256), 0, as_stream(stream), env, env_n, (uint2 *)out, n, g_irr_nphi, g_irr_ntheta, row_begin);
    } else {
        dim3 grid(n, row_end - row_begin);
        hipLaunchKernelGGL(tiled ? irradiance_texel_kernel<true> : irradiance_texel_kernel<false>, grid, dim3(256), 0, as_stream(stream), env, env_n, (uint2 *)out, n, g_irr_nphi, g_irr_ntheta, row_begin);
    }
    return launch_status("vkr_ibl_irradiance");
}

namespace vkr {
static int irr_regions_per_side(int env_n) { return (env_n + (1 << IRR_REGION_SHIFT) - 1) >> IRR_REGION_SHIFT; }
static int64_t irr_partial_floats(int env_n, int n_texels)
{
    const int per_side = irr_regions_per_side(env_n);
    const int64_t n_groups = (n_texels + IRR_GROUP - 1) / IRR_GROUP;
    return (int64_t)6 * per_side * per_side * 4 * n_groups * IRR_GROUP;
}
}  // namespace vkr

extern "C" int64_t vkr_ibl_irradiance_by_region_workspace_bytes(int32_t env_n, int32_t n, int32_t row_begin, int32_t row_end)
{
    using namespace vkr;
    if (env_n <= 0 || env_n > 8192 || n <= 0 || n > 4096 || row_begin < 0 || row_end < row_begin || row_end > 6 * n) {
        fail(VKR_ERR_INVALID, "vkr_ibl_irradiance_by_region_workspace_bytes: bad sizes (env %d, cube %d, rows [%d,%d))", env_n, n, row_begin, row_end);
        return -1;
    }
    return irr_partial_floats(env_n, n * (row_end - row_begin)) * 4 + 256;
}

extern "C" int32_t vkr_ibl_irradiance_by_region(const void *env, int32_t env_n, int32_t env_layout, void *out, int32_t n, int32_t row_begin, int32_t row_end,
                                                void *workspace, int64_t workspace_bytes, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(env && out && env_n > 0, "vkr_ibl_irradiance_by_region: an image is null or env size is bad");
    VKR_REQUIRE(env_n <= 8192 && n <= 4096, "vkr_ibl_irradiance_by_region: sizes beyond 8192 (environment: the kernel addresses the cube with 32-bit byte offsets) or 4096 (irradiance cube): %d, %d", env_n, n);
    int32_t rc = check_rows("vkr_ibl_irradiance_by_region", n, 6 * n, row_begin, row_end);
    if (rc != VKR_OK) return rc;
    if ((rc = check_env_layout("vkr_ibl_irradiance_by_region", env_n, env_layout)) != VKR_OK) return rc;
    const bool tiled = env_layout == VKR_ENV_LAYOUT_TILED_4X4;
    const int n_texels = n * (row_end - row_begin);
    const int64_t floats = irr_partial_floats(env_n, n_texels), need = floats * 4 + 256;
    VKR_REQUIRE(workspace != nullptr && workspace_bytes >= need, "vkr_ibl_irradiance_by_region: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)need);
    VKR_REQUIRE(((uintptr_t)workspace & 3u) == 0, "vkr_ibl_irradiance_by_region: workspace is not 4-byte aligned");
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    if (n_texels == 0) return VKR_OK;
    const int per_side = irr_regions_per_side(env_n), n_regions = 6 * per_side * per_side, n_groups = (n_texels + IRR_GROUP - 1) / IRR_GROUP;
    const int64_t workgroups = (int64_t)((n_regions + 7) / 8) * 8 * n_groups;
    VKR_REQUIRE(workgroups < (1ll << 31), "vkr_ibl_irradiance_by_region: too much work for one call, pass fewer rows");
    float *partial = (float *)workspace;
    uint32_t *uncovered = (uint32_t *)(partial + floats);
    hipStream_t s = as_stream(stream);
    VKR_HIP_CHECK(hipMemsetAsync(uncovered, 0, 4, s));
    hipLaunchKernelGGL(tiled ? irradiance_region_kernel<true> : irradiance_region_kernel<false>, dim3((unsigned)workgroups), dim3(256), 0, s,
                       env, env_n, n, g_irr_nphi, g_irr_ntheta, row_begin, n_texels, partial, per_side, n_regions, n_groups);
    if ((rc = launch_status("vkr_ibl_irradiance_by_region")) != VKR_OK) return rc;
    hipLaunchKernelGGL(irradiance_regions_sum_kernel, dim3((n_texels + 255) / 256), dim3(256), 0, s,
                       partial, n_regions, n_groups, (uint2 *)out, n, g_irr_nphi, g_irr_ntheta, row_begin, n_texels, uncovered);
    return launch_status("vkr_ibl_irradiance_by_region");
}

extern "C" int64_t vkr_env_footprint_tiles_bytes(int32_t n)
{
    using namespace vkr;
    if (n <= 0 || n > 8192) { fail(VKR_ERR_INVALID, "vkr_env_footprint_tiles_bytes: face size %d outside [1, 8192]", n); return -1; }
    const int64_t k = footprint_tiles_per_side(n);
    return 6 * k * k * 128;
}

extern "C" int32_t vkr_tile_rgba16f_footprints(const void *linear, void *tiles, int32_t n, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(linear && tiles, "vkr_tile_rgba16f_footprints: an image is null");
    VKR_REQUIRE(n > 0 && n <= 8192, "vkr_tile_rgba16f_footprints: face size %d outside [1, 8192]", n);
    const int64_t k = footprint_tiles_per_side(n), rows = 6 * k * k * 4;  // one lane per tile row
    const char *a = (const char *)linear, *b = (const char *)tiles;
    VKR_REQUIRE(a + (size_t)6 * n * n * 8 <= b || b + rows * 32 <= a, "vkr_tile_rgba16f_footprints: source and destination overlap");
    hipLaunchKernelGGL(footprint_tiles_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const uint2 *)linear, (uint4 *)tiles, n, (unsigned)k, (unsigned)rows);
    return launch_status("vkr_tile_rgba16f_footprints");
}

namespace vkr {
static int64_t irr_rows_partial_floats(int env_n, int n_texels)
{
    int per_side, cpr;
    irr_region_grid(env_n, &per_side, &cpr);
    const int64_t n_groups = (n_texels + IRR_GROUP - 1) / IRR_GROUP;
    return (int64_t)6 * per_side * per_side * 4 * n_groups * IRR_GROUP;
}
}  // namespace vkr

extern "C" int64_t vkr_ibl_irradiance_by_region_rows_workspace_bytes(int32_t env_n, int32_t n, int32_t row_begin, int32_t row_end)
{
    using namespace vkr;
    if (env_n <= 0 || env_n > 8192 || n <= 0 || n > 4096 || row_begin < 0 || row_end < row_begin || row_end > 6 * n) {
        fail(VKR_ERR_INVALID, "vkr_ibl_irradiance_by_region_rows_workspace_bytes: bad sizes (env %d, cube %d, rows [%d,%d))", env_n, n, row_begin, row_end);
        return -1;
    }
    return irr_rows_partial_floats(env_n, n * (row_end - row_begin)) * 4 + 256;
}

extern "C" int32_t vkr_ibl_irradiance_by_region_rows(const void *env_tiles, int32_t env_n, void *out, int32_t n, int32_t row_begin, int32_t row_end,
                                                     void *workspace, int64_t workspace_bytes, void *stream)
{
    using namespace vkr;
    const char *who = "vkr_ibl_irradiance_by_region_rows";
    VKR_REQUIRE(env_tiles && out && env_n > 0, "%s: an image is null or env size is bad", who);
    VKR_REQUIRE(env_n <= 8192 && n <= 4096, "%s: sizes beyond 8192 (environment) or 4096 (irradiance cube): %d, %d", who, env_n, n);
    int32_t rc = check_rows(who, n, 6 * n, row_begin, row_end);
    if (rc != VKR_OK) return rc;
    const int n_texels = n * (row_end - row_begin);
    const int64_t floats = irr_rows_partial_floats(env_n, n_texels), need = floats * 4 + 256;
    VKR_REQUIRE(workspace != nullptr && workspace_bytes >= need, "%s: workspace of %lld bytes, %lld needed", who, (long long)workspace_bytes, (long long)need);
    VKR_REQUIRE(((uintptr_t)workspace & 3u) == 0 && ((uintptr_t)env_tiles & 7u) == 0, "%s: workspace is not 4-byte aligned or the environment not 8-byte aligned", who);
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    if (n_texels == 0) return VKR_OK;
    int per_side, cpr;
    irr_region_grid(env_n, &per_side, &cpr);
    const int n_regions = 6 * per_side * per_side, n_groups = (n_texels + IRR_GROUP - 1) / IRR_GROUP;
    const int64_t workgroups = (int64_t)((n_regions + 7) / 8) * 8 * n_groups;
    VKR_REQUIRE(workgroups < (1ll << 31), "%s: too much work for one call, pass fewer rows", who);
    float *partial = (float *)workspace;
    uint32_t *uncovered = (uint32_t *)(partial + floats);
    hipStream_t s = as_stream(stream);
    VKR_HIP_CHECK(hipMemsetAsync(uncovered, 0, 4, s));
    hipLaunchKernelGGL(irradiance_rows_kernel, dim3((unsigned)workgroups), dim3(256), 0, s,
                       (const char *)env_tiles, env_n, (unsigned)footprint_tiles_per_side(env_n), n, g_irr_nphi, g_irr_ntheta, row_begin, n_texels, partial, per_side, cpr, n_regions, n_groups);
    if ((rc = launch_status(who)) != VKR_OK) return rc;
    hipLaunchKernelGGL(irradiance_regions_sum_kernel, dim3((n_texels + 255) / 256), dim3(256), 0, s,
                       partial, n_regions, n_groups, (uint2 *)out, n, g_irr_nphi, g_irr_ntheta, row_begin, n_texels, uncovered);
    return launch_status(who);
}

namespace vkr {
// Launches the prefilter kernel over `m` (one or more mips) with `group` texels per workgroup.
static void launch_prefilter(const void *env, int env_n, int env_layout, PrefilterMips &m, int group, hipStream_t s)
{
    int per_xcd_total = 0;
    for (int i = 0; i < m.count; i++) {
        m.per_xcd[i] = ((m.n_texels[i] + group - 1) / group + 7) / 8;  // groups of texels per XCD
        per_xcd_total += m.per_xcd[i];
    }
    auto kernel = group == 16 ? (env_layout == VKR_ENV_LAYOUT_FOOTPRINT_TILES ? prefilter_texel_kernel<VKR_ENV_LAYOUT_FOOTPRINT_TILES, 16>
                                 : env_layout == VKR_ENV_LAYOUT_TILED_4X4 ? prefilter_texel_kernel<VKR_ENV_LAYOUT_TILED_4X4, 16> : prefilter_texel_kernel<VKR_ENV_LAYOUT_LINEAR, 16>)
                             : (env_layout == VKR_ENV_LAYOUT_FOOTPRINT_TILES ? prefilter_texel_kernel<VKR_ENV_LAYOUT_FOOTPRINT_TILES, 2>
                                 : env_layout == VKR_ENV_LAYOUT_TILED_4X4 ? prefilter_texel_kernel<VKR_ENV_LAYOUT_TILED_4X4, 2> : prefilter_texel_kernel<VKR_ENV_LAYOUT_LINEAR, 2>);
    hipLaunchKernelGGL(kernel, dim3(8u * (unsigned)per_xcd_total), dim3(256), 0, s, env, env_n, m);
}

static int32_t check_prefilter_env(const char *who, const void *env, int env_n, int env_layout)
{
    VKR_REQUIRE(env && env_n > 0, "%s: the environment is null or its size is bad", who);
    if (env_layout == VKR_ENV_LAYOUT_FOOTPRINT_TILES) {
        VKR_REQUIRE(env_n <= 8192 && ((uintptr_t)env & 7u) == 0, "%s: footprint tiles need a face size of at most 8192 and 8-byte alignment", who);
        return VKR_OK;
    }
    return check_env_layout(who, env_n, env_layout);
}
}  // namespace vkr

extern "C" int32_t vkr_ibl_prefilter(const void *env, int32_t env_n, int32_t env_layout, void *out, int32_t n, float roughness, int32_t row_begin, int32_t row_end, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(out != nullptr, "vkr_ibl_prefilter: the output image is null");
    VKR_REQUIRE(roughness >= 0.0f && roughness <= 1.0f, "vkr_ibl_prefilter: roughness %g outside [0,1]", (double)roughness);
    int32_t rc = check_prefilter_env("vkr_ibl_prefilter", env, env_n, env_layout);
    if (rc != VKR_OK) return rc;
    if ((rc = check_rows("vkr_ibl_prefilter", n, 6 * n, row_begin, row_end)) != VKR_OK) return rc;
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    if (row_end == row_begin) return VKR_OK;
    PrefilterMips m = {};
    m.count = 1; m.out[0] = (uint2 *)out; m.n[0] = n; m.rough[0] = roughness; m.row_begin[0] = row_begin; m.n_texels[0] = n * (row_end - row_begin);
    launch_prefilter(env, env_n, env_layout, m, m.n_texels[0] >= 65536 ? 16 : 2, as_stream(stream));
    return launch_status("vkr_ibl_prefilter");
}

extern "C" int32_t vkr_ibl_prefilter_mips(const VkrPrefilterMipsArgs *a, void *stream)
{
    using namespace vkr;
    const char *who = "vkr_ibl_prefilter_mips";
    VKR_REQUIRE(a != nullptr, "%s: args is null", who);
    VKR_REQUIRE(a->mips >= 1 && a->mips <= VKR_PREFILTER_MAX_MIPS, "%s: %d mips (1 .. %d)", who, a->mips, VKR_PREFILTER_MAX_MIPS);
    VKR_REQUIRE(a->base_res > 0 && (a->base_res >> (a->mips - 1)) >= 1, "%s: a base size of %d has no %d mips", who, a->base_res, a->mips);
    int32_t rc = check_prefilter_env(who, a->env, a->env_res, a->env_layout);
    if (rc != VKR_OK) return rc;
    for (int m = 0; m < a->mips; m++) {
        const int n = a->base_res >> m;
        VKR_REQUIRE(a->out[m] != nullptr, "%s: the image of mip %d is null", who, m);
        if ((rc = check_rows(who, n, 6 * n, a->row_begin[m], a->row_end[m])) != VKR_OK) return rc;
    }
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    // a mip with many texels is bound by instruction issue and takes sixteen texels per workgroup in a launch of its own;
    // the others -- bound by cache-line traffic, and too small to fill the chip one by one -- share one launch
    PrefilterMips rest = {};
    for (int m = 0; m < a->mips; m++) {
        const int n = a->base_res >> m, texels = n * (a->row_end[m] - a->row_begin[m]);
        if (texels == 0) continue;
        const float rough = a->mips > 1 ? (float)m / (float)(a->mips - 1) : 0.0f;  // VulkanRenderer.cs:874
        PrefilterMips one = {};
        PrefilterMips &dst = texels >= 65536 ? one : rest;
        const int k = dst.count++;
        dst.out[k] = (uint2 *)a->out[m]; dst.n[k] = n; dst.rough[k] = rough; dst.row_begin[k] = a->row_begin[m]; dst.n_texels[k] = texels;
        if (&dst == &one) launch_prefilter(a->env, a->env_res, a->env_layout, one, 16, as_stream(stream));
    }
    if (rest.count) launch_prefilter(a->env, a->env_res, a->env_layout, rest, 2, as_stream(stream));
    return launch_status(who);
}

extern "C" int32_t vkr_ibl_brdf_lut(void *out, int32_t n, int32_t row_begin, int32_t row_end, void *stream)
{
    using namespace vkr;
    VKR_REQUIRE(out != nullptr, "vkr_ibl_brdf_lut: output image is null");
    int32_t rc = check_rows("vkr_ibl_brdf_lut", n, n, row_begin, row_end);
    if (rc != VKR_OK) return rc;
    if ((rc = ensure_tables()) != VKR_OK) return rc;
    if (row_end == row_begin) return VKR_OK;
    dim3 grid((n + 255) / 256, row_end - row_begin);
    hipLaunchKernelGGL(brdf_lut_kernel, grid, dim3(256), 0, as_stream(stream), (uint32_t *)out, n, row_begin, row_end);
    return launch_status("vkr_ibl_brdf_lut");
}
