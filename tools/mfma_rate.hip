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

// the tap-GEMM's register pattern: 4 weight float4, 4 pixel float4, 16 accumulators, MFMA (s, n, m) order
__global__ __launch_bounds__(512) void probe_tile(unsigned long long* out, float* sink, int iters, int slot) {
    f32x4 acc[4][4], w[4], x[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        w[n] = f32x4{threadIdx.x * 0.001f + n, 1.f + n, 2.f + n, 3.f + n};
        x[n] = f32x4{1.01f + n, 0.5f + n, threadIdx.x * 0.002f, 0.25f + n};
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0, 0, 0, 0};
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][s4], x[m][s4], acc[n][m], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(acc[n][m]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = t1 - t0;
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) sum += acc[n][m][m];
    sink[blockIdx.x * 512 + threadIdx.x] = sum;
}

// the tap-GEMM main loop, feature by feature: F & 1 = 8 streaming dwordx4 loads per 64-MFMA step into the other register
// set, F & 2 = the 16 v_cndmask of the padding mask, F & 4 = per-step LDS table read (uint4 + u32) for the load offsets
template <int F>
__global__ __launch_bounds__(256, 2) void probe_loop(unsigned long long* out, float* sink, int iters, int slot, const float* buf, unsigned mask, int nwin) {
    __shared__ uint4 tab[4][64];
    __shared__ unsigned tok[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tab[wave][lane] = make_uint4(lane * 4, lane * 4 + 256, lane * 4 + 512, lane * 4 + 768);
    tok[wave][lane] = mask;
    f32x4 acc[4][4];
    struct St { f32x4 w[4], x[4]; unsigned ok; } A, B;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        A.w[n] = f32x4{threadIdx.x * 0.001f + n, 1.f + n, 2.f + n, 3.f + n};
        A.x[n] = f32x4{1.01f + n, 0.5f + n, threadIdx.x * 0.002f, 0.25f + n};
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0, 0, 0, 0};
    }
    A.ok = mask; B = A;
    const float* p = buf + ((size_t)((blockIdx.x * 4 + wave) % nwin)) * 65536;   // 256 KB windows; nwin sets the footprint
    unsigned ofs = 0;
    auto issue = [&](St& S) {
        uint4 o = make_uint4(lane * 4, lane * 4 + 256, lane * 4 + 512, lane * 4 + 768);
        unsigned ok = mask;
        if constexpr (F & 4) { o = tab[wave][lane]; ok = tok[wave][lane]; }
        if constexpr (F & 1) {
#pragma unroll
            for (int n = 0; n < 4; ++n) S.w[n] = *reinterpret_cast<const f32x4*>(p + ((ofs + n * 1024 + lane * 4) & 65535));
            S.x[0] = *reinterpret_cast<const f32x4*>(p + ((ofs + 4096 + o.x) & 65535));
            S.x[1] = *reinterpret_cast<const f32x4*>(p + ((ofs + 4096 + o.y) & 65535));
            S.x[2] = *reinterpret_cast<const f32x4*>(p + ((ofs + 4096 + o.z) & 65535));
            S.x[3] = *reinterpret_cast<const f32x4*>(p + ((ofs + 4096 + o.w) & 65535));
            ofs += 8192;
        }
        S.ok = ok;
    };
    auto finish = [&](St& S) {
        if constexpr (F & 2) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bool in = (S.ok >> m) & 1u;
                S.x[m].x = in ? S.x[m].x : 0.f; S.x[m].y = in ? S.x[m].y : 0.f; S.x[m].z = in ? S.x[m].z : 0.f; S.x[m].w = in ? S.x[m].w : 0.f;
            }
        }
    };
    auto mma = [&](const St& S, int s4) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(S.w[n][s4], S.x[m][s4], acc[n][m], 0, 0, 0);
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    issue(A);
    for (int it = 0; it < iters / 4; ++it) {            // 128 MFMAs per iteration, as the kernel's step pair
        finish(A); mma(A, 0); issue(B); mma(A, 1); mma(A, 2); mma(A, 3);
        if constexpr (F & 8) {      // scheduling hint: one VMEM read after every 8 MFMAs instead of a burst of 8
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        finish(B); mma(B, 0); issue(A); mma(B, 1); mma(B, 2); mma(B, 3);
        if constexpr (F & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(acc[n][m]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = t1 - t0;
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) sum += acc[n][m][m];
    sink[blockIdx.x * 256 + threadIdx.x] = sum;
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
        {"16x16x4 f32, tap-GEMM tile pattern, 1 wave/SIMD", probe_tile, 256}, {"16x16x4 f32, tap-GEMM tile pattern, 2 waves/SIMD", probe_tile, 512},
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
    {
        float* buf;
        hipMalloc(&buf, (size_t)512 * 4 * 65536 * 4);
        hipMemset(buf, 0, (size_t)512 * 4 * 65536 * 4);
        float* sink2; hipMalloc(&sink2, 512 * 256 * 4);
        typedef void (*LoopFn)(unsigned long long*, float*, int, int, const float*, unsigned, int);
        const LoopFn fns[] = {probe_loop<0>, probe_loop<1>, probe_loop<3>, probe_loop<7>, probe_loop<2>, probe_loop<9>};
        const char* nm[] = {"MFMAs only", "+ 8 streaming loads / step", "+ loads + 16 cndmask", "+ loads + cndmask + LDS table", "+ 16 cndmask only", "+ 8 loads, one per 8 MFMAs"};
        const int wins[] = {2048, 128, 16, 1};          // 512 MB (HBM), 32 MB (Infinity Cache), 4 MB (L2), 256 KB
        for (int wi = 0; wi < 4; ++wi)
        for (int c = 0; c < 6; ++c) {
            if (wi > 0 && (c == 0 || c == 4)) continue;
            float msl = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(fns[c], dim3(512), dim3(256), 0, 0, out, sink2, 1024, 15, buf, 15u, wins[wi]);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&msl, e0, e1);
            }
            // 512 workgroups x 4 waves on 1024 SIMDs = 2 waves per SIMD, 1024 * 32 MFMAs per wave
            printf("tap-GEMM loop, 2 waves/SIMD, footprint %4d MB, %-32s %.2f ns per MFMA per SIMD\n", wins[wi] / 4, nm[c], msl * 1e6 / (2.0 * 1024 * 32));
        }
    }
    for (int c = 0; c < n; ++c)
        printf("%-40s %.1f s_memtime ticks per MFMA of one wave | launch %.1f us = %.1f ns per MFMA of one wave | %.2f ticks/ns\n", cases[c].name,
               (double)h[c] / (iters * 32), ms[c] * 1e3, ms[c] * 1e6 / (iters * 32), (double)h[c] / (ms[c] * 1e6));
    return 0;
}
