// Correctly rounded reciprocal and square root built from the hardware's 1-ulp estimates
// (v_rcp_f32, v_rsq_f32) and fused multiply-adds.  They return the same bits as IEEE 1/x and
// sqrt(x) on these domains: fast_rcp_rn for 2^-126 <= |x| < 2^126 (x and 1/x both normal),
// fast_sqrt_rn for 2^-96 <= x <= 2^96; vkr_selftest_math() checks both exhaustively on the device.
// They cost 3 and 5 instructions instead of the ~11 and ~14 of the compiler's IEEE expansions, and
// apart from the estimate only multiplies and fmas, which issue faster than compares and selects.
#pragma once
#include <hip/hip_runtime.h>

#define VKR_FM_DEV __device__ __forceinline__

namespace vkr {

// RN(1/x): hardware estimate, one Newton step evaluated with fma (Markstein).
VKR_FM_DEV float fast_rcp_rn(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// Two Newton steps (kept for the self-test: shows whether the second step changes anything).
VKR_FM_DEV float fast_rcp_rn2(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// RN(1/x) for every x: the fast sequence inside its exact domain, the compiler's IEEE division
// outside it (zero, subnormal, huge, inf, NaN).
VKR_FM_DEV float rcp_rn_checked(float x)
{
    float ax = __builtin_fabsf(x);
    if (__builtin_expect(!(ax >= 0x1p-126f && ax < 0x1p126f), 0)) return 1.0f / x;
    return fast_rcp_rn(x);
}

// RN(sqrt(x)): y = hardware estimate of 1/sqrt(x), g = x*y a root within an ulp or so, then one
// correction with the exact residual x - g*g (Markstein's final step).  No input in the domain
// rounds the wrong way (swept exhaustively).
VKR_FM_DEV float fast_sqrt_rn(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float r = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(r, h, g);
}

// RN(sqrt(x)) for every x: zero returns itself, everything else outside [2^-96, 2^96] takes the
// compiler's IEEE sequence.
VKR_FM_DEV float sqrt_rn_checked(float x)
{
    if (__builtin_expect(!(x >= 0x1p-96f && x <= 0x1p96f), 0)) return x == 0.0f ? x : __builtin_sqrtf(x);
    return fast_sqrt_rn(x);
}

// s = RN(sqrt(x)) and RN(1/s): the pair normalize() and length() need.  Outside
// [2^-96, 2^96] (where the fma residuals above could lose bits, and for 0 / inf / NaN) it falls
// back to the compiler's IEEE sequences, so the pair equals sqrtf(x), 1.0f/sqrtf(x) for every x.
VKR_FM_DEV void sqrt_and_rcp_rn(float x, float &s, float &inv)
{
    if (__builtin_expect(!(x >= 0x1p-96f && x <= 0x1p96f), 0)) {
        s = __builtin_sqrtf(x);
        inv = 1.0f / s;
    } else {
        s = fast_sqrt_rn(x);
        inv = fast_rcp_rn(s);
    }
}

}  // namespace vkr
