// What does "BatchNorm finalise inside the producing launch" cost on this chip?  (candidates)
// A stand-in for a convolution's epilogue: G workgroups each write one partial row [2][C] floats, then
//   form A: a second, dependent launch (C / 4 workgroups) sums the rows in fp64 and writes the per-channel result -- what
//           bn_finalize_fwd_kernel does today;
//   form B: ONE launch -- rows by device-scope (sc1) stores, `s_waitcnt vmcnt(0)`, an agent-scope ticket; the last-arriving
//           workgroup sums all rows with sc1 loads (fixed order) and writes the result, resets the ticket.
// Both forms are followed by a dependent consumer launch (reads the result), as bn_act / the next convolution would be, and the
// triple is timed back to back: (form A) - (form B) is what fusing saves per BatchNorm.
//   hipcc -O3 --offload-arch=gfx950 tools/ticket_tail.hip -o tools/ticket_tail && ./tools/ticket_tail
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((vector_size(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000); }
constexpr int SC1 = 16;      // cache-policy bit of the buffer intrinsics: device scope

// the "convolution": spin ~work_us of ALU time, then write this workgroup's row (threads 0 .. 2C/4 - 1, 16 bytes each)
template <bool FUSED>
__global__ __launch_bounds__(256) void producer(float* rows, int C, long spin, unsigned* ticket, float* result) {
    float v = (float)threadIdx.x;
    for (long i = 0; i < spin; ++i) v = v * 1.0000001f + 0.5f;
    const int nq = 2 * C / 4;
    const __amdgpu_buffer_rsrc_t rr = rsrc(rows);
    if ((int)threadIdx.x < nq) {
        f32x4 r = {v * 1e-30f + 1.f, 2.f, 3.f, 4.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, r), rr, (int)(((long)blockIdx.x * nq + threadIdx.x) * 16), 0, FUSED ? SC1 : 0);
    }
    if constexpr (FUSED) {
        __shared__ unsigned last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        __syncthreads();
        if (!last) return;
        // the last workgroup: thread t -> channel quad t % (C/4), row group t / (C/4); fp64; fixed order
        const int Q = C / 4, RG = 256 / Q, q = threadIdx.x % Q, rg = threadIdx.x / Q;
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        // (eight rows = sixteen loads in flight per thread; rows beyond the grid read row 0 and are masked)
        for (int r0 = rg; r0 < (int)gridDim.x; r0 += 8 * RG) {
            f32x4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * RG < (int)gridDim.x ? r0 + u * RG : 0;
                a[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(((long)r * nq + q) * 16), 0, SC1));
                b[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(((long)r * nq + Q + q) * 16), 0, SC1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + u * RG < (int)gridDim.x)
                    for (int e = 0; e < 4; ++e) { s1[e] += a[u][e]; s2[e] += b[u][e]; }
        }
        __shared__ double sm[256][8];
        for (int e = 0; e < 4; ++e) { sm[threadIdx.x][e] = s1[e]; sm[threadIdx.x][4 + e] = s2[e]; }
        __syncthreads();
        if ((int)threadIdx.x < Q) {
            double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < RG; ++k) for (int e = 0; e < 8; ++e) t[e] += sm[k * Q + threadIdx.x][e];
            for (int e = 0; e < 4; ++e) { result[threadIdx.x * 4 + e] = (float)(t[e] / gridDim.x); result[C + threadIdx.x * 4 + e] = (float)(1.0 / sqrt(t[4 + e] + 1e-3)); }
        }
        if (threadIdx.x == 0) *ticket = 0u;
    }
}
__global__ __launch_bounds__(256) void finalize(const float* rows, int nrows, int C, float* result) {
    const int c0 = blockIdx.x * 4, nq = 2 * C / 4;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int r = threadIdx.x; r < nrows; r += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(rows + ((long)r * nq + blockIdx.x) * 4), b = *reinterpret_cast<const f32x4*>(rows + ((long)r * nq + C / 4 + blockIdx.x) * 4);
        for (int e = 0; e < 4; ++e) { s1[e] += a[e]; s2[e] += b[e]; }
    }
    __shared__ double sm[4][8];
    for (int e = 0; e < 4; ++e) for (int o = 32; o > 0; o >>= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    if ((threadIdx.x & 63) == 0) for (int e = 0; e < 4; ++e) { sm[threadIdx.x >> 6][e] = s1[e]; sm[threadIdx.x >> 6][4 + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int e = threadIdx.x;
        result[c0 + e] = (float)((sm[0][e] + sm[1][e] + sm[2][e] + sm[3][e]) / nrows);
        result[C + c0 + e] = (float)(1.0 / sqrt(sm[0][4 + e] + sm[1][4 + e] + sm[2][4 + e] + sm[3][4 + e] + 1e-3));
    }
}
__global__ __launch_bounds__(256) void consumer(const float* result, float* out, int C, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = result[i % C] * 2.f + result[C + i % C];
}

int main() {
    float *rows, *result, *out;
    unsigned* ticket;
    hipMalloc(&rows, 8192L * 2 * 128 * 4); hipMalloc(&result, 2 * 128 * 4); hipMalloc(&out, 1 << 24); hipMalloc(&ticket, 4);
    hipMemset(ticket, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int G, C; const char* what; } cfgs[] = {{1024, 64, "64 ch, 256 x 512 x 32 (1024 tiles)"}, {256, 128, "128 ch (256 tiles)"}, {4096, 16, "16 ch (4096 tiles)"},
                                                          {3200, 64, "64 ch, config 3 (3200 tiles)"}, {512, 64, "64 ch, one row per workgroup (512)"}};
    for (const Cfg& c : cfgs) {
        float ms[2];
        for (int form = 0; form < 2; ++form) {
            const int iters = 200;
            for (int it = -20; it < iters; ++it) {
                if (it == 0) hipEventRecord(e0, 0);
                if (form == 0) {
                    hipLaunchKernelGGL(producer<false>, dim3(c.G), dim3(256), 0, 0, rows, c.C, 2000L, ticket, result);
                    hipLaunchKernelGGL(finalize, dim3(c.C / 4), dim3(256), 0, 0, rows, c.G, c.C, result);
                } else {
                    hipLaunchKernelGGL(producer<true>, dim3(c.G), dim3(256), 0, 0, rows, c.C, 2000L, ticket, result);
                }
                hipLaunchKernelGGL(consumer, dim3(1024), dim3(256), 0, 0, result, out, c.C, 1L << 20);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[form], e0, e1);
            ms[form] = ms[form] / iters * 1e3f;
        }
        float h[2];
        hipMemcpy(h, result, 8, hipMemcpyDeviceToHost);
        printf("%-40s producer + finalize kernel + consumer %6.2f us | fused (ticket, last workgroup) + consumer %6.2f us | saved %5.2f us   (result %.3f)\n",
               c.what, ms[0], ms[1], ms[0] - ms[1], h[0]);
    }
    return 0;
}
