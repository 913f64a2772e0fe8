// What does ONE vector-memory instruction cost the fp32 MFMA stream of its SIMD?  Item = 64 v_mfma_f32_32x32x2_f32 per wave (the
// tap-GEMM's K-step); behind the first 16 MFMAs the wave issues V loads of kind K on L1-resident addresses (16 KB footprint per
// workgroup), waits for them at the end of the item.  512 threads = 2 waves per SIMD, no barriers (the shipped kernel's occupancy).
//   kinds: global_load_dwordx4 (64-bit VGPR address) | saddr form (SGPR base + 32-bit VGPR offset) | buffer_load_dwordx4 offen |
//          global_load_lds_dwordx4 | buffer_load_dwordx4 ... lds | global_load_dwordx2 | global_load_dword
// hipcc -O3 --offload-arch=gfx950 tools/vmem_cost.hip -o tools/vmem_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) float lds[];

template <int K, int V>
__global__ __launch_bounds__(512, 1) void vm(float* sink, const float* src, int items) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 A0 = f32x4{0.1f + lane, 0.2f, 0.3f, 0.4f}, A1 = A0 * 1.5f, B0 = A0 * 0.7f, B1 = A0 * 0.3f;
    const float* base = src + blockIdx.x * 4096;                    // 16 KB per workgroup
    const unsigned off0 = (unsigned)(lane * 16 + wave * 1024) & 16383u;
    i32x4 rsrc;
    {
        const unsigned long long b = (unsigned long long)base;
        rsrc[0] = (int)(unsigned)b; rsrc[1] = (int)(unsigned)(b >> 32); rsrc[2] = 16384; rsrc[3] = 0x00020000;
        for (int q = 0; q < 4; ++q) rsrc[q] = __builtin_amdgcn_readfirstlane(rsrc[q]);
    }
    float dummy = 0.f;
    unsigned long long dummy64 = 0;
    const float* fixedp[16];
    for (int v = 0; v < 16; ++v) { fixedp[v] = base + ((off0 + v * 2048u) & 16383u) / 4; asm volatile("" : "+v"(fixedp[v])); }
    __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 16384, 0x00020000);
    for (int it = 0; it < items; ++it) {
        f32x4 ld[V > 0 ? V : 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s], B0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s], B1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], B0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], B1[s], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const unsigned off = (off0 + (unsigned)v * 2048u + (unsigned)(it & 1) * 8192u) & 16383u;
                    if constexpr (K == 0) {
                        const float* p = base + off / 4;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[v]) : "v"(p) : "memory");
                    } else if constexpr (K == 1) {
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld[v]) : "v"(off), "s"(base) : "memory");
                    } else if constexpr (K == 2) {
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[v]) : "v"(off), "s"(rsrc) : "memory");
                    } else if constexpr (K == 3) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off / 4),
                                                         (__attribute__((address_space(3))) void*)(lds + (wave * V + v) * 256), 16, 0, 0);
                    } else if constexpr (K == 4) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(prsrc, (__attribute__((address_space(3))) void*)(lds + (wave * V + v) * 256),
                                                                 16, (int)off, 0, 0, 0);
                    } else if constexpr (K == 7) {
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[v]) : "v"(fixedp[v]) : "memory");
                    } else if constexpr (K == 8) {
                        unsigned long long t64 = (unsigned long long)base + off;
                        asm volatile("" : "+v"(t64));
                        dummy64 ^= t64;
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld[v]) : "v"(off), "s"(base) : "memory");
                    } else if constexpr (K == 9) {
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ld[v]) : "v"(fixedp[0]), "n"(v * 256) : "memory");
                    } else if constexpr (K == 5) {
                        f32x2 t;
                        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(base + off / 4) : "memory");
                        ld[v] = f32x4{t.x, t.y, 0.f, 0.f};
                    } else if constexpr (K == 6) {
                        float t;
                        asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(base + off / 4) : "memory");
                        ld[v] = f32x4{t, 0.f, 0.f, 0.f};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (V > 0 && K != 3 && K != 4) {
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("" :: "v"(ld[v]));
        }
    }
    float sum = dummy + (float)(dummy64 & 1);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][lane & 15];
    sink[blockIdx.x * 512 + threadIdx.x] = sum + lds[threadIdx.x];
}

static double g_base_ms = 0.0;
template <int K, int V>
void run(const char* name, float* sink, const float* src) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vm<K, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int items = 20000;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((vm<K, V>), dim3(256), dim3(512), 65536, 0, sink, src, items);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    if (V == 0) g_base_ms = ms;
    const double tf = 2048.0 * items * 64.0 * 4096.0 / (ms * 1e-3) / 1e12;
    // extra wall time per VMEM instruction and SIMD (2 waves per SIMD each issue V per item)
    const double ns = V ? (ms - g_base_ms) * 1e6 / ((double)items * V * 2) : 0.0;
    printf("%-58s %8.2f ms  %7.1f TFLOP/s   +%.1f ns of matrix issue per VMEM instruction\n", name, ms, tf, ns);
    fflush(stdout);
}

int main() {
    float *sink, *src; (void)hipMalloc(&sink, 512 * 512 * 4); (void)hipMalloc(&src, 256 * 16384 + 65536); (void)hipMemset(src, 0, 256 * 16384 + 65536);
    run<0, 0>("MFMAs only (2 waves/SIMD)", sink, src);
    run<0, 8>("8 x global_load_dwordx4 (VGPR address pair)", sink, src);
    run<1, 8>("8 x global_load_dwordx4 (SGPR base + VGPR offset)", sink, src);
    run<2, 8>("8 x buffer_load_dwordx4 offen", sink, src);
    run<3, 8>("8 x global_load_lds_dwordx4", sink, src);
    run<4, 8>("8 x buffer_load_dwordx4 offen lds", sink, src);
    run<7, 8>("8 x global_load_dwordx4, loop-invariant VGPR address pairs", sink, src);
    run<9, 8>("8 x global_load_dwordx4, ONE VGPR pair + immediate offsets", sink, src);
    run<8, 8>("8 x saddr form + 8 64-bit VALU adds (address math only)", sink, src);
    run<5, 8>("8 x global_load_dwordx2", sink, src);
    run<6, 8>("8 x global_load_dword", sink, src);
    run<5, 16>("16 x global_load_dwordx2", sink, src);
    run<0, 4>("4 x global_load_dwordx4", sink, src);
    run<0, 16>("16 x global_load_dwordx4", sink, src);
    return 0;
}
