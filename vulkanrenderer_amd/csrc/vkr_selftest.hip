// Device self-test: sweeps every binary32 bit pattern and counts where the fast correctly rounded
// reciprocal / square root of vkr_fastmath.h differ from the compiler's IEEE 1/x and sqrt(x).
#include "vkr_fastmath.h"
#include "vkr_host.h"

namespace vkr {

// counts[0] rcp (1 Newton step)   counts[1] rcp (2 steps)   counts[2] sqrt on x >= 2^-96
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
        if (ex >= 31 && __builtin_bit_cast(unsigned int, s) != __builtin_bit_cast(unsigned int, sw)) local[2]++;  // x >= 2^-96
        if (__builtin_bit_cast(unsigned int, fast_rcp_rn(sw)) != __builtin_bit_cast(unsigned int, iw)) local[3]++;
        float s2, inv2;
        sqrt_and_rcp_rn(x, s2, inv2);
        if (__builtin_bit_cast(unsigned int, inv2) != __builtin_bit_cast(unsigned int, iw) ||
            __builtin_bit_cast(unsigned int, s2) != __builtin_bit_cast(unsigned int, sw)) local[4]++;
    }
    for (int k = 0; k < 7; k++)
        if (local[k]) atomicAdd(&counts[k], local[k]);
}

}  // namespace vkr

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
