// Device self-test: sweeps every binary32 bit pattern and counts where the fast correctly rounded
// reciprocal / square root of vkr_fastmath.h differ from the compiler's IEEE 1/x and sqrt(x).
#include "vkr_fastmath.h"
#include "vkr_host.h"

namespace vkr {

// counts[0] rcp (1 Newton step)   counts[1] rcp (2 steps)   counts[2] sqrt on 2^-96 <= x <= 2^96
// counts[3] rcp applied to IEEE sqrt(x)   counts[4] the sqrt + reciprocal pair as shipped
// counts[5] patterns tested for rcp   counts[6] patterns tested for sqrt
__global__ __launch_bounds__(256) void selftest_math_kernel(unsigned long long *counts)
{
    unsigned long long local[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long total = 1ull << 32;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < total; b += (unsigned long long)gridDim.x * 256) {
        unsigned int bits = (unsigned int)b;
        float x = __builtin_bit_cast(float, bits);
        unsigned int ex = (bits >> 23) & 0xFF;
        if (ex < 1 || ex > 252) continue;  // normal x whose reciprocal is also normal: 2^-126 <= |x| < 2^126
        float want = 1.0f / x;
        local[5]++;
        if (__builtin_bit_cast(unsigned int, fast_rcp_rn(x)) != __builtin_bit_cast(unsigned int, want)) local[0]++;
        if (__builtin_bit_cast(unsigned int, fast_rcp_rn2(x)) != __builtin_bit_cast(unsigned int, want)) local[1]++;
        if (bits >> 31) continue;
        local[6]++;
        float sw = __builtin_sqrtf(x), iw = 1.0f / sw;
        float s = fast_sqrt_rn(x);
        if (x >= 0x1p-96f && x <= 0x1p96f && __builtin_bit_cast(unsigned int, s) != __builtin_bit_cast(unsigned int, sw)) local[2]++;
        if (__builtin_bit_cast(unsigned int, fast_rcp_rn(sw)) != __builtin_bit_cast(unsigned int, iw)) local[3]++;
        float s2, inv2;
        sqrt_and_rcp_rn(x, s2, inv2);
        if (__builtin_bit_cast(unsigned int, inv2) != __builtin_bit_cast(unsigned int, iw) ||
            __builtin_bit_cast(unsigned int, s2) != __builtin_bit_cast(unsigned int, sw)) local[4]++;
    }
    for (int k = 0; k < 7; k++)
        if (local[k]) atomicAdd(&counts[k], local[k]);
}

// 8 independent fma chains per lane, 4096 fmas each: the FP32 vector rate a kernel can reach.
__global__ __launch_bounds__(256) void peak_fma_kernel(float *out, float a, float b)
{
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (float)(threadIdx.x + k);
    for (int it = 0; it < 4096; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_fmaf(v[k], a, b);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 16 bytes per lane in, 16 out
__global__ __launch_bounds__(256) void peak_copy_kernel(const uint4 *src, uint4 *dst, size_t n)
{
    // four independent 16-byte loads per lane before the first store
    const size_t i = ((size_t)blockIdx.x * 256 * 4) + threadIdx.x;
    if (i + 3 * 256 < n) {
        const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
}

}  // namespace vkr

extern "C" int32_t vkr_measure_peaks(float *fp32_tflops, float *copy_gbs)
{
    using namespace vkr;
    VKR_REQUIRE(fp32_tflops && copy_gbs, "vkr_measure_peaks: an output is null");
    const size_t copy_bytes = (size_t)1 << 30;  // 1 GiB each way: well past the 256 MiB last-level cache
    const int fma_blocks = 256 * 16;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int32_t rc = VKR_OK;
    auto body = [&]() -> int32_t {
        VKR_HIP_CHECK(hipMalloc(&a, copy_bytes));
        VKR_HIP_CHECK(hipMalloc(&b, copy_bytes));
        VKR_HIP_CHECK(hipMemset(a, 1, copy_bytes));
        VKR_HIP_CHECK(hipEventCreate(&e0));
        VKR_HIP_CHECK(hipEventCreate(&e1));
        float ms = 0.0f;
        for (int pass = 0; pass < 2; pass++) {  // the first pass warms up
            VKR_HIP_CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(peak_fma_kernel, dim3(fma_blocks), dim3(256), 0, nullptr, (float *)b, 0.999f, 0.001f);
            VKR_HIP_CHECK(hipEventRecord(e1, nullptr));
            VKR_HIP_CHECK(hipEventSynchronize(e1));
            VKR_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        *fp32_tflops = (float)((double)fma_blocks * 256 * 8 * 4096 * 2 / (ms * 1e-3) / 1e12);
        for (int pass = 0; pass < 2; pass++) {
            VKR_HIP_CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(peak_copy_kernel, dim3((unsigned)(copy_bytes / 16 / 1024)), dim3(256), 0, nullptr, (const uint4 *)a, (uint4 *)b, copy_bytes / 16);
            VKR_HIP_CHECK(hipEventRecord(e1, nullptr));
            VKR_HIP_CHECK(hipEventSynchronize(e1));
            VKR_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        *copy_gbs = (float)(2.0 * (double)copy_bytes / (ms * 1e-3) / 1e9);
        return launch_status("vkr_measure_peaks");
    };
    rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

extern "C" int32_t vkr_selftest_math(uint64_t *counts7)
{
    using namespace vkr;
    VKR_REQUIRE(counts7 != nullptr, "vkr_selftest_math: output is null");
    unsigned long long *dev = nullptr;
    VKR_HIP_CHECK(hipMalloc((void **)&dev, 7 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(dev, 0, 7 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_math_kernel, dim3(256 * 32), dim3(256), 0, nullptr, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(counts7, dev, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(VKR_ERR_HIP, "vkr_selftest_math: %s", hipGetErrorString(e));
    return VKR_OK;
}
