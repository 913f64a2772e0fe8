// Shared host/device helpers for liblanefit_hip.so (gfx950 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lanefit.h"

#define LF_WAVE 64

extern thread_local char lf_err_buf[512];
int lf_fail(const char* fmt, ...);

#define LF_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) return lf_fail(__VA_ARGS__); \
    } while (0)

#define LF_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) return lf_fail("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int lf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// wave64 all-lanes sum (butterfly), double and float
__device__ __forceinline__ double lf_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float lf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
