// Host-side helpers shared by the translation units of libvkr_hip.so: error reporting and the
// HIP status check.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vkr_hip.h"

namespace vkr {

// Records a message for vkr_last_error() and returns `code`.
int32_t fail(int32_t code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define VKR_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return vkr::fail(VKR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                                 \
    } while (0)

#define VKR_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return vkr::fail(VKR_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// Stores `value` into n consecutive 64-bit words of device memory (one RGBA16F texel or two D32F texels each).
int32_t fill_u64(void *dst, size_t n, unsigned long long value, void *stream);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the status of the launch that was just issued.
inline int32_t launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VKR_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
    return VKR_OK;
}

}  // namespace vkr
