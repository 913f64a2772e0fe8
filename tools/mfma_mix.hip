// What keeps a wave's fp32 MFMA stream below the 154 TFLOP/s a bare stream reaches (tools/mfma_sustain.hip)?  One "item" = 64
// v_mfma_f32_32x32x2_f32 on a 64 x 64 wave tile (4 accumulators of 16 registers) = the LDS-tiled tap-GEMM's compute phase.
//   F & 1: the item's 16 ds_read_b128 fragment reads (4 per 16 MFMAs, prefetched one group ahead)
//   F & 2: the 8 v_cndmask per group (padding mask)
//   F & 4: ping-pong -- 512 threads, waves 0-3 and 4-7 alternate compute / idle phases through s_barrier
//   F & 8: 512 threads, both waves of a SIMD compute concurrently, one s_barrier per item
//   otherwise 256 threads (a lone wave per SIMD), no barrier
// hipcc -O3 --offload-arch=gfx950 tools/mfma_mix.hip -o tools/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern __shared__ __attribute__((aligned(16))) float lds[];

template <int F>
__global__ __launch_bounds__(512, 2) void mix(float* sink, int items, unsigned vb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
    __syncthreads();
    const int swl = (lane >> 1) & 7, lh = lane >> 5, l31 = lane & 31;
    int fo[4];
    for (int q = 0; q < 4; ++q) fo[q] = ((2 * q + lh) ^ swl) * 4;
    const float* pa = lds + (l31 + (wave & 1) * 64) * 32;
    const float* pb = lds + 8192 + (l31 + ((wave >> 1) & 1) * 64) * 32;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 A0[2], A1[2], B0[2], B1[2];
    auto frag = [&](int q, int s) {
        if constexpr (F & 1) {
            A0[s] = *reinterpret_cast<const f32x4*>(pa + fo[q]); A1[s] = *reinterpret_cast<const f32x4*>(pa + 1024 + fo[q]);
            B0[s] = *reinterpret_cast<const f32x4*>(pb + fo[q]); B1[s] = *reinterpret_cast<const f32x4*>(pb + 1024 + fo[q]);
        }
    };
    for (int s = 0; s < 2; ++s) { A0[s] = f32x4{0.1f + lane, 0.2f, 0.3f, 0.4f}; A1[s] = A0[s] * 1.5f; B0[s] = A0[s] * 0.7f; B1[s] = A0[s] * 0.3f; }
    const bool v0 = (vb >> (lane & 1)) & 1u, v1 = (vb >> 1) & 1u;
    auto compute = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = q & 1;
            if (q < 3) frag(q + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 b0 = B0[cb], b1 = B1[cb];
            if constexpr (F & 2) {
                b0.x = v0 ? b0.x : 0.f; b0.y = v0 ? b0.y : 0.f; b0.z = v0 ? b0.z : 0.f; b0.w = v0 ? b0.w : 0.f;
                b1.x = v1 ? b1.x : 0.f; b1.y = v1 ? b1.y : 0.f; b1.z = v1 ? b1.z : 0.f; b1.w = v1 ? b1.w : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cb][s], b0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cb][s], b1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb][s], b0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb][s], b1[s], acc[1][1], 0, 0, 0);
            }
        }
    };
    if constexpr (F & 4) {
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < items; ++it) {
            frag(0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else {
        for (int it = 0; it < items; ++it) {
            frag(0, 0);
            compute();
            if constexpr (F & 8) __builtin_amdgcn_s_barrier();
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][lane & 15];
    sink[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int F>
void run(const char* name, float* sink, int threads) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int items = 20000;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mix<F>, dim3(256), dim3(threads), 65536, 0, sink, items, 3u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double waves = 256.0 * threads / 64;
    const double tf = waves * items * 64.0 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
    printf("%-64s %8.2f ms  %7.1f TFLOP/s\n", name, ms, tf);
    fflush(stdout);
}

int main() {
    float* sink; (void)hipMalloc(&sink, 512 * 512 * 4);
    run<0>("lone wave, MFMAs only", sink, 256);
    run<1>("lone wave, + fragment reads", sink, 256);
    run<3>("lone wave, + fragment reads + masks", sink, 256);
    run<8>("2 waves/SIMD concurrent, barrier per item, MFMAs only", sink, 512);
    run<9>("2 waves/SIMD concurrent, + fragment reads", sink, 512);
    run<11>("2 waves/SIMD concurrent, + fragment reads + masks", sink, 512);
    run<4>("ping-pong, MFMAs only", sink, 512);
    run<5>("ping-pong, + fragment reads", sink, 512);
    run<7>("ping-pong, + fragment reads + masks", sink, 512);
    return 0;
}
