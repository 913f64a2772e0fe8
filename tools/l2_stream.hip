// How fast can a CU pull the bf16 tap-GEMM's pixel operand out of L2 / Infinity Cache, by access pattern?  No matrix work, no weights:
// persistent workgroups of 4 waves walk 256-pixel tiles of an (NPIX x 128 channel) bf16 tensor (256 B per pixel), three taps per
// tile (pixel offsets -8, 0, +8), every wave loading its 64 pixels x 256 B per tap into registers, double-buffered by tap, and
// XOR-folding them; optionally storing 64 pixels x 256 B per tile like the epilogue does (8 B per lane).
//   PAT 0: instruction = 16 pixels x 64 B  (lane -> pixel l&15, 16-byte block l>>4: the MFMA operand layout), the four 64-byte
//          quarters of a pixel by consecutive instructions                                     (tapgemm_bf16_stream_kernel)
//   PAT 1: same instruction shape, but quarter-major: all 64 pixels' first quarter, then the second ...   (the LDS rings' K-steps)
//   PAT 2: instruction = 8 pixels x 128 B  (lane -> pixel l>>3, block l&7): whole lines per instruction
//   PAT 3: instruction = 4 pixels x 256 B  (lane -> pixel l>>4, block l&15): whole pixels per instruction
// STORE 1 / 3 / 2: the bf16 epilogue's 8-byte stores / the fp32 epilogue's 16-byte stores of 16 pixels x 64 B / whole lines
// hipcc -O3 --offload-arch=gfx950 tools/l2_stream.hip -o tools/l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((vector_size(16)));
typedef unsigned u32x2v __attribute__((vector_size(8)));

template <int PAT, int STORE>
__global__ __launch_bounds__(256, 2) void pull(const void* x, void* y, unsigned* sink, int npix, int taps) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (unsigned)npix * 256u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, (unsigned)npix * 256u, 0x00020000);
    const int ntiles = npix / 256;
    unsigned first = blockIdx.x, stride = gridDim.x, end = (unsigned)ntiles;
    if ((gridDim.x & 7u) == 0 && (ntiles & 7) == 0) {
        const unsigned per = (unsigned)ntiles >> 3;
        first = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); stride = gridDim.x >> 3; end = (blockIdx.x & 7u) * per + per;
    }
    // instruction i (0..15) of a tap -> byte offset inside the wave's 64 pixels x 256 B
    unsigned lo[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (PAT == 0) lo[i] = (unsigned)(((i >> 2) * 16 + (lane & 15)) * 256 + (i & 3) * 64 + (lane >> 4) * 16);
        else if (PAT == 1) lo[i] = (unsigned)(((i & 3) * 16 + (lane & 15)) * 256 + (i >> 2) * 64 + (lane >> 4) * 16);
        else if (PAT == 2) lo[i] = (unsigned)(((i >> 1) * 8 + (lane >> 3)) * 256 + (i & 1) * 128 + (lane & 7) * 16);
        else lo[i] = (unsigned)((i * 4 + (lane >> 4)) * 256 + (lane & 15) * 16);
    }
    u32x4v A[16], B[16];
    u32x4 acc = {0u, 0u, 0u, 0u};
    auto issue = [&](u32x4v (&S)[16], unsigned tile, int t) __attribute__((always_inline)) {
        const int p0 = (int)(tile * 256u + wave * 64) + (t - 1) * 8;          // first pixel of the wave at this tap
        const unsigned base = (unsigned)(p0 < 0 ? 0 : p0) * 256u;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(base + lo[i]), 0, 0);
    };
    auto fold = [&](const u32x4v (&S)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.x ^= S[i][0]; acc.y ^= S[i][1]; acc.z ^= S[i][2]; acc.w ^= S[i][3]; }
    };
    if (first < end) issue(A, first, 0);
    for (unsigned tile = first; tile < end; tile += stride) {
        // taps is odd: pairs A, B then the last in A; B receives the first tap of the next tile
        int t = 0;
        for (; t + 1 < taps; t += 2) {
            issue(B, tile, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            fold(A);
            __builtin_amdgcn_sched_barrier(0);
            issue(A, tile, t + 2 < taps ? t + 2 : 0);
            __builtin_amdgcn_sched_barrier(0);
            fold(B);
            __builtin_amdgcn_sched_barrier(0);
        }
        issue(B, tile + stride < end ? tile + stride : tile, 0);
        __builtin_amdgcn_sched_barrier(0);
        fold(A);
        if (STORE == 2) {
            // whole lines: lane -> pixel l>>3, 16-byte block l&7 (8 pixels x 128 B per instruction), 16 instructions
            const unsigned ob = (tile * 256u + wave * 64) * 256u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u32x4v v = {acc.x + (unsigned)i, acc.y, acc.z, acc.w};
                __builtin_amdgcn_raw_buffer_store_b128(v, ry, (int)(ob + (unsigned)(((i >> 1) * 8 + (lane >> 3)) * 256 + (i & 1) * 128 + (lane & 7) * 16)), 0, 0);
            }
        }
        if (STORE == 3) {
            // the fp32 epilogue's store shape: lane (pl, kq) writes 16 bytes (4 fp32 channels) of pixel m*16 + pl, channel tile n:
            // 64 bytes of each of 16 lines per instruction
            const unsigned ob = (tile * 256u + wave * 64) * 256u;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    u32x4v v = {acc.x + (unsigned)m, acc.y + (unsigned)n, acc.z, acc.w};
                    __builtin_amdgcn_raw_buffer_store_b128(v, ry, (int)(ob + (unsigned)((m * 16 + (lane & 15)) * 256 + n * 64 + (lane >> 4) * 16)), 0, 0);
                }
        }
        if (STORE == 1) {
            // the epilogue's store shape: lane (pl, kq) writes 8 bytes (4 bf16 channels) of pixel m*16 + pl, channel tile n
            const unsigned ob = (tile * 256u + wave * 64) * 256u;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    u32x2v v = {acc.x + (unsigned)m, acc.y + (unsigned)n};
                    __builtin_amdgcn_raw_buffer_store_b64(v, ry, (int)(ob + (unsigned)((m * 16 + (lane & 15)) * 256 + n * 32 + (lane >> 4) * 8)), 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) A[i] = B[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

template <int PAT, int STORE>
void run(const char* name, const void* x, void* y, unsigned* sink, int npix, int grid, int taps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((pull<PAT, STORE>), dim3(grid), dim3(256), 0, 0, x, y, sink, npix, taps);
    hipEventRecord(a);
    const int iters = 50;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pull<PAT, STORE>), dim3(grid), dim3(256), 0, 0, x, y, sink, npix, taps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, rd = (double)npix * 256 * taps, wr = STORE ? (double)npix * 256 : 0.;
    printf("%-58s grid %4d taps %d | %7.1f us | operand reads %6.2f TB/s (%5.1f MB) | tensor bytes (1 read%s) %5.2f TB/s\n", name, grid, taps, us,
           rd / us / 1e6, rd / 1e6, STORE ? " + 1 write" : "", ((double)npix * 256 + wr) / us / 1e6);
}

int main(int argc, char** argv) {
    const int npix = argc > 1 ? atoi(argv[1]) : 204800;
    void *x, *y;
    unsigned* sink;
    hipMalloc(&x, (size_t)npix * 256); hipMalloc(&y, (size_t)npix * 256); hipMalloc(&sink, 64);
    hipMemset(x, 1, (size_t)npix * 256);
    for (int grid : {512}) {
        for (int taps : {1, 3}) {
            run<0, 0>("16 px x 64 B per instruction, pixel quarters consecutive", x, y, sink, npix, grid, taps);
            run<1, 0>("16 px x 64 B per instruction, quarter-major", x, y, sink, npix, grid, taps);
            run<2, 0>("8 px x 128 B per instruction", x, y, sink, npix, grid, taps);
            run<3, 0>("4 px x 256 B per instruction", x, y, sink, npix, grid, taps);
        }
        run<0, 1>("16 px x 64 B, consecutive + the epilogue's 8-byte stores", x, y, sink, npix, grid, 3);
        run<3, 1>("4 px x 256 B + the epilogue's 8-byte stores", x, y, sink, npix, grid, 3);
        run<0, 2>("16 px x 64 B, consecutive + whole-line 16-byte stores", x, y, sink, npix, grid, 3);
        run<3, 2>("4 px x 256 B + whole-line 16-byte stores", x, y, sink, npix, grid, 3);
        run<2, 2>("8 px x 128 B + whole-line 16-byte stores", x, y, sink, npix, grid, 3);
        run<0, 3>("16 px x 64 B, consecutive + 16-byte stores of 16 px x 64 B (the fp32 epilogue)", x, y, sink, npix, grid, 3);
        run<0, 3>("16 px x 64 B, one tap + 16-byte stores of 16 px x 64 B", x, y, sink, npix, grid, 1);
        run<0, 2>("16 px x 64 B, one tap + whole-line 16-byte stores", x, y, sink, npix, grid, 1);
    }
    return 0;
}
