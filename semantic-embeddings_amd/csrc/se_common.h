// se_common.h -- shared helpers for the gfx950 kernels behind include/sehip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/sehip.h"

namespace se {

// thread-local last-error text (se_last_error)
char *err_buf();
int fail(int code, const char *fmt, ...);

#define SE_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return se::fail(SE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                               \
    } while (0)

#define SE_LAUNCH_CHECK()                                                                      \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess)                                                                 \
            return se::fail(SE_ERR_HIP, "kernel launch failed: %s (%s:%d)",                    \
                            hipGetErrorString(e__), __FILE__, __LINE__);                       \
    } while (0)

constexpr int WAVE = 64;

// Tuning / test switches (environment variables that pin a kernel variant, print phase profiles or make a kernel skip work)
// exist only in the -DSE_TUNING build (sehip/libsehip_tuning.so, loaded by tests / tools through SEHIP_LIB): the product
// library ignores them all, so no environment variable can change what an entry point computes.  Product switches that stay:
// SE_RANK_SAFE=1 (guaranteed-order ranking kernel) and SE_RANK_VERBOSE=1 (says which ranking kernel the probe selected).
#ifdef SE_TUNING
inline const char *tuning_env(const char *name) { return getenv(name); }
constexpr bool kTuning = true;
#else
inline const char *tuning_env(const char *) { return nullptr; }
constexpr bool kTuning = false;
#endif

// Order-preserving float32 -> uint32 key of the canonical ranking order:
// ascending value, -0.0 == +0.0, every NaN maps to 0xFFFFFFFF (sorted last).
__device__ __forceinline__ uint32_t canon_key(float f)
{
    uint32_t u = __float_as_uint(f);
    if (f != f) return 0xFFFFFFFFu;
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even float32 -> bf16 (NaN preserved)
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if (f != f) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace se
