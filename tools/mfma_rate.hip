// Issue-rate probe (gfx950): back-to-back MFMAs of one wave per SIMD on 4 independent accumulators, cycles by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(1.0f + i * 0.01f); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
    const float af = threadIdx.x * 0.001f, bf = 1.01f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[q], 0, 0, 0);
                else if constexpr (KIND == 1) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[q], 0, 0, 0);
                else acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[q], 0, 0, 0);
            }
    }
    asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 256 * 4);
    const int iters = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, sink, iters);
        hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, sink, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[3];
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    const char* names[3] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16", "v_mfma_f32_16x16x4_f32"};
    for (int k = 0; k < 3; ++k) printf("%s: %.1f s_memtime ticks per MFMA (one wave per SIMD, 4 independent accumulators)\n", names[k], (double)h[k] / (iters * 32));
    return 0;
}
