// Shared device/host helpers for libpyprob_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pyprob_amd.h"

namespace pp {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);

#define PP_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            pp::set_error(__VA_ARGS__);  \
            return PP_EINVAL;            \
        }                                \
    } while (0)

#define PP_LAUNCH_CHECK(name)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            pp::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                    \
        }                                                                       \
    } while (0)

#define PP_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__ != 0) return rc__; \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 64-lane butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace pp
