// Sustained matrix-core rate (gfx950): is the fp32 MFMA roof power-limited?  Every SIMD runs back-to-back MFMAs (1 or 2 waves per
// SIMD) for 1 ms ... 0.5 s; TFLOP/s by HIP events.  hipcc -O3 --offload-arch=gfx950 tools/mfma_sustain.hip -o tools/mfma_sustain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void burn(float* sink, long iters) {
    const float af = threadIdx.x * 0.001f, bf = 1.01f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(1.0f + i * 0.01f); }
    float sum = 0.f;
    if constexpr (KIND == 0) {          // v_mfma_f32_32x32x2_f32, 4 accumulators
        f32x16 acc[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[q], 0, 0, 0);
        for (int q = 0; q < 4; ++q) sum += acc[q][q];
    } else if constexpr (KIND == 3) {   // v_mfma_f32_32x32x2_f32 on random operands (8 A and 8 B values per lane, cycled)
        f32x16 acc[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        float ar[8], br[8];
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; ar[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
            h = h * 1664525u + 1013904223u; br[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        }
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[r], br[(r + q) & 7], acc[q], 0, 0, 0);
        for (int q = 0; q < 4; ++q) sum += acc[q][q];
    } else if constexpr (KIND == 1) {   // v_mfma_f32_16x16x4_f32, 16 accumulators
        f32x4 acc[16];
        for (int q = 0; q < 16; ++q) acc[q] = f32x4{0, 0, 0, 0};
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[q], 0, 0, 0);
        for (int q = 0; q < 16; ++q) sum += acc[q][q & 3];
    } else {                            // v_mfma_f32_16x16x32_bf16, 16 accumulators
        f32x4 acc[16];
        for (int q = 0; q < 16; ++q) acc[q] = f32x4{0, 0, 0, 0};
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[q], 0, 0, 0);
        for (int q = 0; q < 16; ++q) sum += acc[q][q & 3];
    }
    sink[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
    float* sink; hipMalloc(&sink, 512 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"f32 32x32x2", "f32 16x16x4", "bf16 16x16x32", "f32 32x32x2 rnd"};
    const double flop_per_mfma[4] = {2.0 * 32 * 32 * 2, 2.0 * 16 * 16 * 4, 2.0 * 16 * 16 * 32, 2.0 * 32 * 32 * 2};
    const int mfma_per_iter[4] = {32, 64, 64, 32};
    for (int kind = 3; kind >= 0; kind -= 3)
        for (int threads = 256; threads <= 512; threads += 256)
            for (long iters = 10000; iters <= 1000000; iters *= 10) {
                float ms = 0.f;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0, 0);
                    if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(256), dim3(threads), 0, 0, sink, iters);
                    else if (kind == 1) hipLaunchKernelGGL(burn<1>, dim3(256), dim3(threads), 0, 0, sink, iters);
                    else if (kind == 3) hipLaunchKernelGGL(burn<3>, dim3(256), dim3(threads), 0, 0, sink, iters);
                    else hipLaunchKernelGGL(burn<2>, dim3(256), dim3(threads), 0, 0, sink, iters);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double waves = 256.0 * threads / 64;
                const double tf = waves * iters * mfma_per_iter[kind] * flop_per_mfma[kind] / (ms * 1e-3) / 1e12;
                printf("%-14s %d waves/SIMD  %8.2f ms  %8.1f TFLOP/s\n", names[kind], threads / 256, ms, tf);
                fflush(stdout);
            }
    return 0;
}
