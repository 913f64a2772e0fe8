// Storage-typed 4-channel vector access shared by the bandwidth-bound kernels and the tap-GEMM epilogues.
// Activations and gradient buffers are fp32 (precision modes 0 and 1) or bf16 (mode 2); arithmetic is fp32 in
// both: a bf16 value widens exactly (bits << 16), results are rounded to nearest even on store
// (v_cvt_pk_bf16_f32).  Per-channel vectors (BN scale/shift, bias, dropout masks, statistics) are always fp32.
#pragma once
#include <hip/hip_runtime.h>

typedef float lf_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lf_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lf_bf16;

template <typename T>
__device__ __forceinline__ lf_f32x4 lf_ldv(const T* p);
template <>
__device__ __forceinline__ lf_f32x4 lf_ldv<float>(const float* p) { return *reinterpret_cast<const lf_f32x4*>(p); }
template <>
__device__ __forceinline__ lf_f32x4 lf_ldv<lf_bf16>(const lf_bf16* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    lf_f32x4 v;
    v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
    v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
    return v;
}
template <typename T>
__device__ __forceinline__ void lf_stv(T* p, lf_f32x4 v);
template <>
__device__ __forceinline__ void lf_stv<float>(float* p, lf_f32x4 v) { *reinterpret_cast<lf_f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void lf_stv<lf_bf16>(lf_bf16* p, lf_f32x4 v) {
    lf_bf16x4 b;
    b[0] = (lf_bf16)v.x; b[1] = (lf_bf16)v.y; b[2] = (lf_bf16)v.z; b[3] = (lf_bf16)v.w;
    *reinterpret_cast<lf_bf16x4*>(p) = b;
}
// scalar forms
template <typename T>
__device__ __forceinline__ float lf_ld1(const T* p) { return (float)*p; }
template <typename T>
__device__ __forceinline__ void lf_st1(T* p, float v) { *p = (T)v; }

// NQ channel quads per thread in one access: fp32 one quad (16 bytes), bf16 two (8 channels = 16 bytes: an 8-byte access per lane
// touches a cache line for half of what a wave instruction can take from it and ran the bf16 BatchNorm passes at ~0.7x the fp32
// kernels' bandwidth)
template <typename T, int NQ>
__device__ __forceinline__ void lf_ldq(const T* p, lf_f32x4 (&v)[NQ]) {
    if constexpr (NQ == 1) v[0] = lf_ldv(p);
    else {
        static_assert(NQ == 2 && sizeof(T) == 2, "two quads per access: bf16 only");
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        v[0].x = __uint_as_float(r.x << 16); v[0].y = __uint_as_float(r.x & 0xffff0000u);
        v[0].z = __uint_as_float(r.y << 16); v[0].w = __uint_as_float(r.y & 0xffff0000u);
        v[1].x = __uint_as_float(r.z << 16); v[1].y = __uint_as_float(r.z & 0xffff0000u);
        v[1].z = __uint_as_float(r.w << 16); v[1].w = __uint_as_float(r.w & 0xffff0000u);
    }
}
template <typename T, int NQ>
__device__ __forceinline__ void lf_stq(T* p, const lf_f32x4 (&v)[NQ]) {
    if constexpr (NQ == 1) lf_stv(p, v[0]);
    else {
        static_assert(NQ == 2 && sizeof(T) == 2, "two quads per access: bf16 only");
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t b;
        b[0] = (lf_bf16)v[0].x; b[1] = (lf_bf16)v[0].y; b[2] = (lf_bf16)v[0].z; b[3] = (lf_bf16)v[0].w;
        b[4] = (lf_bf16)v[1].x; b[5] = (lf_bf16)v[1].y; b[6] = (lf_bf16)v[1].z; b[7] = (lf_bf16)v[1].w;
        *reinterpret_cast<bf16x8_t*>(p) = b;
    }
}
