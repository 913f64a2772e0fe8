// Issue-rate probe (gfx950): back-to-back MFMAs of one wave per SIMD on 4 independent accumulators, cycles by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NACC = 4>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float* sink, int iters, int slot) {
    f32x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = f32x4{0, 0, 0, 0};
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(1.0f + i * 0.01f); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
    const float af = threadIdx.x * 0.001f, bf = 1.01f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                if constexpr (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[q], 0, 0, 0);
                else if constexpr (KIND == 1) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[q], 0, 0, 0);
                else acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[q], 0, 0, 0);
            }
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) asm volatile("" ::"v"(acc[q]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = t1 - t0;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q) sum += acc[q][q & 3];
    sink[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256); hipMalloc(&sink, 256 * 512 * 4);
    const int iters = 256;
    struct Case { const char* name; void (*fn)(unsigned long long*, float*, int, int); int threads; };
    const Case cases[] = {
        {"16x16x32 bf16, 4 acc, 1 wave/SIMD", probe<0, 4>, 256},  {"16x16x32 bf16, 8 acc, 1 wave/SIMD", probe<0, 8>, 256},
        {"16x16x32 bf16, 16 acc, 1 wave/SIMD", probe<0, 16>, 256}, {"16x16x32 bf16, 4 acc, 2 waves/SIMD", probe<0, 4>, 512},
        {"16x16x32 bf16, 16 acc, 2 waves/SIMD", probe<0, 16>, 512}, {"16x16x16 bf16, 4 acc, 1 wave/SIMD", probe<1, 4>, 256},
        {"16x16x4 f32, 4 acc, 1 wave/SIMD", probe<2, 4>, 256},    {"16x16x4 f32, 16 acc, 2 waves/SIMD", probe<2, 16>, 512},
    };
    const int n = sizeof(cases) / sizeof(cases[0]);
    float ms[16];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        for (int c = 0; c < n; ++c) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(cases[c].fn, dim3(256), dim3(cases[c].threads), 0, 0, out, sink, iters, c);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[c], e0, e1);
        }
    }
    unsigned long long h[16];
    hipMemcpy(h, out, n * 8, hipMemcpyDeviceToHost);
    // per-wave ticks per MFMA; HIP-event time of the whole launch (one workgroup per CU) calibrates the tick
    for (int c = 0; c < n; ++c)
        printf("%-40s %.1f s_memtime ticks per MFMA of one wave | launch %.1f us = %.1f ns per MFMA of one wave | %.2f ticks/ns\n", cases[c].name,
               (double)h[c] / (iters * 32), ms[c] * 1e3, ms[c] * 1e6 / (iters * 32), (double)h[c] / (ms[c] * 1e6));
    return 0;
}
