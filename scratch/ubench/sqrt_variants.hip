// Exhaustive check of FP32-only sequences for RN(sqrt(x)) and RN(1/RN(sqrt(x))) on 2^-96 <= x <= 2^96
// (candidates to replace the integer-neighbour selection of fast_sqrt_rn).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned bits(float f) { return __builtin_bit_cast(unsigned, f); }

__global__ __launch_bounds__(256) void sweep(unsigned long long *counts)
{
    unsigned long long local[8] = {};
    const unsigned lo = 0x0F800000u /* 2^-96 */, hi = 0x6F800000u /* 2^96 */;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * 256 + threadIdx.x; b <= hi; b += (unsigned long long)gridDim.x * 256) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        const float sw = __builtin_sqrtf(x), iw = 1.0f / sw;
        local[7]++;
        {   // A: hw sqrt, hw rcp of it, one fma correction each
            const float s0 = __builtin_amdgcn_sqrtf(x), r0 = __builtin_amdgcn_rcpf(s0);
            const float e = __builtin_fmaf(-s0, s0, x);
            const float s = __builtin_fmaf(e, 0.5f * r0, s0);
            const float e2 = __builtin_fmaf(-s, r0, 1.0f);
            const float inv = __builtin_fmaf(e2, r0, r0);
            if (bits(s) != bits(sw)) local[0]++;
            if (bits(inv) != bits(iw)) local[1]++;
        }
        {   // B: rsq based (Markstein): g = x*y, h = y/2, one correction; reciprocal from y
            const float y = __builtin_amdgcn_rsqf(x);
            const float g = x * y, h = 0.5f * y;
            const float r = __builtin_fmaf(-g, g, x);
            const float s = __builtin_fmaf(r, h, g);
            const float e2 = __builtin_fmaf(-s, y, 1.0f);
            const float inv = __builtin_fmaf(e2, y, y);
            if (bits(s) != bits(sw)) local[2]++;
            if (bits(inv) != bits(iw)) local[3]++;
        }
        {   // C: as A, but the reciprocal gets a second Newton step
            const float s0 = __builtin_amdgcn_sqrtf(x), r0 = __builtin_amdgcn_rcpf(s0);
            const float e = __builtin_fmaf(-s0, s0, x);
            const float s = __builtin_fmaf(e, 0.5f * r0, s0);
            float e2 = __builtin_fmaf(-s, r0, 1.0f);
            float inv = __builtin_fmaf(e2, r0, r0);
            e2 = __builtin_fmaf(-s, inv, 1.0f);
            inv = __builtin_fmaf(e2, inv, inv);
            if (bits(inv) != bits(iw)) local[4]++;
        }
        {   // E: B's root; reciprocal by two Newton steps from the rsq estimate
            const float y = __builtin_amdgcn_rsqf(x);
            const float g = x * y, h = 0.5f * y;
            const float r = __builtin_fmaf(-g, g, x);
            const float s = __builtin_fmaf(r, h, g);
            float e2 = __builtin_fmaf(-s, y, 1.0f);
            float inv = __builtin_fmaf(e2, y, y);
            e2 = __builtin_fmaf(-s, inv, 1.0f);
            inv = __builtin_fmaf(e2, inv, inv);
            if (bits(inv) != bits(iw)) local[6]++;
        }
        {   // D: as A, two corrections of the root
            const float s0 = __builtin_amdgcn_sqrtf(x), r0 = __builtin_amdgcn_rcpf(s0);
            const float hh = 0.5f * r0;
            float e = __builtin_fmaf(-s0, s0, x);
            float s = __builtin_fmaf(e, hh, s0);
            e = __builtin_fmaf(-s, s, x);
            s = __builtin_fmaf(e, hh, s);
            if (bits(s) != bits(sw)) local[5]++;
        }
    }
    for (int k = 0; k < 8; k++)
        if (local[k]) atomicAdd(&counts[k], local[k]);
}

int main()
{
    unsigned long long *dev, host[8];
    CHECK(hipMalloc(&dev, sizeof(host)));
    CHECK(hipMemset(dev, 0, sizeof(host)));
    hipLaunchKernelGGL(sweep, dim3(256 * 32), dim3(256), 0, 0, dev);
    CHECK(hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost));
    printf("tested %llu\nA sqrt %llu  A rcp %llu\nB sqrt %llu  B rcp %llu\nC rcp %llu\nD sqrt %llu\nE rcp %llu\n", host[7], host[0], host[1], host[2], host[3], host[4], host[5], host[6]);
    return 0;
}
