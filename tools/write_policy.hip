// Why do the forward launches of the bandwidth-bound layers run slower than backward launches that move MORE bytes (fp32 16-channel
// convolution: 134 MB forward in 36 us, 201 MB data gradient in 31 us; bf16 64-channel at config 3: 210 MB in 68 us, 315 MB in 60 us)?
// Forward writes every activation into memory nobody touched in this step (a 4.8 GB workspace); backward writes into three ping-pong
// gradient buffers.  This probe is a layer chain without arithmetic: persistent workgroups copy `bytes` from src to dst in whole
// lines (16 B per lane), dst(k) = src(k+1), with
//   dst pattern  ring R: the chain walks R buffers round robin (R = 2: ping-pong, resident in the 256 MB Infinity Cache at 64-105 MB;
//                R = 24: every write goes to lines that left the caches long ago, as in forward)
//   store policy 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt   (global_store_dwordx4 cache-coherence bits)
// Output: us per copy and GB/s (read + write) per (bytes, ring, policy).
//   hipcc -O3 --offload-arch=gfx950 tools/write_policy.hip -o tools/write_policy && tools/write_policy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4v __attribute__((vector_size(16)));

template <int POL>
__device__ __forceinline__ void st16(u32x4v v, char* p) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// a workgroup walks 16 KB chunks (4 x 1 KB instructions per wave) inside its XCD's contiguous range; 4 loads in flight per lane
template <int POL>
__global__ __launch_bounds__(256) void copy_kernel(const char* __restrict__ src, char* __restrict__ dst, unsigned nchunks) {
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const unsigned lo = (unsigned)(((unsigned long long)xcd * nchunks) >> 3), hi = (unsigned)(((unsigned long long)(xcd + 1) * nchunks) >> 3);
    for (unsigned c = lo + slot; c < hi; c += per) {
        const size_t off = (size_t)c * 16384 + threadIdx.x * 16;
        u32x4v v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4v*>(src + off + i * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) st16<POL>(v[i], dst + off + i * 4096);
    }
}

template <int POL>
float run(char* arena, size_t bytes, int ring, int iters, hipStream_t st) {
    const unsigned nchunks = (unsigned)(bytes / 16384);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int k = 0;
    for (int it = 0; it < ring + 2; ++it, ++k)        // warm-up: one lap
        hipLaunchKernelGGL(copy_kernel<POL>, dim3(2048), dim3(256), 0, st, arena + (size_t)(k % ring) * bytes, arena + (size_t)((k + 1) % ring) * bytes, nchunks);
    hipEventRecord(a, st);
    for (int it = 0; it < iters; ++it, ++k)
        hipLaunchKernelGGL(copy_kernel<POL>, dim3(2048), dim3(256), 0, st, arena + (size_t)(k % ring) * bytes, arena + (size_t)((k + 1) % ring) * bytes, nchunks);
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return 1e3f * ms / iters;
}

int main() {
    const size_t sizes[2] = {(size_t)64 << 20, (size_t)100 << 20};
    const int rings[3] = {2, 3, 24};
    char* arena = nullptr;
    if (hipMalloc(&arena, (size_t)24 * (100 << 20)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(arena, 1, (size_t)24 * (100 << 20));
    hipStream_t st;
    hipStreamCreate(&st);
    const char* names[5] = {"default", "nt", "sc1", "sc0 sc1", "sc1 nt"};
    for (size_t bytes : sizes)
        for (int ring : rings) {
            printf("%3zu MB per tensor, chain over %2d buffers |", bytes >> 20, ring);
            for (int pol = 0; pol < 5; ++pol) {
                float us = 0.f;
                if (pol == 0) us = run<0>(arena, bytes, ring, 48, st);
                if (pol == 1) us = run<1>(arena, bytes, ring, 48, st);
                if (pol == 2) us = run<2>(arena, bytes, ring, 48, st);
                if (pol == 3) us = run<3>(arena, bytes, ring, 48, st);
                if (pol == 4) us = run<4>(arena, bytes, ring, 48, st);
                printf("  %s %6.1f us (%4.0f GB/s)", names[pol], us, 2.0 * bytes / us / 1e3);
            }
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
