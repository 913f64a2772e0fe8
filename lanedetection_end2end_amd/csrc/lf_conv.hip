// Tap-GEMM convolution kernels for gfx950 (see lf_conv.h for the contract).
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32, exact fp32, D = A*B + C, wave64):
//   A[i][k]: lane l supplies row i = l&15, k-slot kq = l>>4      -> weights / x-channels
//   B[k][j]: lane l supplies col j = l&15, k-slot kq = l>>4      -> pixels / g-channels
//   D[i][j]: lane l holds rows 4*(l>>4)+e (e = 0..3), col l&15
// Forward/dgrad: rows = output channels, cols = 16 pixels, k = source channels.  Each lane
// loads ONE float4 = 4 consecutive channels (16*cg + 4*kq + s, s = 0..3) of its pixel / of its
// weight row and feeds MFMA step s with element s: the k order is permuted identically on both
// operands, which a contraction does not care about.  A pixel's 16 channels (64 B) are read by 4
// lanes, a 16-pixel tile by one wave instruction; accumulators come out as 4 consecutive output
// channels per lane -> one float4 store per (tile, lane).  No LDS: operands stream from L1/L2
// straight into VGPRs, one (tap, 16-channel) step prefetched ahead of the MFMAs that use it.
// fp32 MFMA runs at the vector rate (157 TF), 1/16 of bf16, so one dwordx4 per operand tile per
// 16 MFMAs (512 cycles) is far below what the load path sustains.
//
// Kernels in this file:
//   tapgemm_kernel            fp32 matrix cores, operands streamed L2 -> VGPR: the 64- / 128-channel launches of the network
//   tapgemm_lean_kernel       16-channel layers in fp32 (HBM-bound)
//   tapgemm_split_kernel      precision mode fp32x9: fp32 results from exact 3-way bf16 splits on the bf16 matrix cores
//   tapgemm_bf16_kernel       precision mode bf16, operands streamed into registers (operand prologue, ragged widths)
//   tapgemm_bf16_wl_kernel    bf16 tensors, the 3-tap convolutions at 64 / 128 channels: memory touched in whole 128-byte lines
//                             (LDS-DMA operand ring, weights in registers, LDS-transposed stores)
//   tapgemm_bf16_ring_kernel  bf16 tensors, every other tap table: persistent LDS-DMA ring
//   tapgemm_bf16_lean_kernel  bf16 tensors, 16 -> 16 channels (two taps per K = 32 MFMA)
//   tapwgrad_kernel / tapwgrad16_kernel / tapwgrad16_tr_kernel   weight gradients, split-K over pixels
//   the split-K reductions (one per weight gradient, or batched per backward pass) and the weight-packing kernels.
// The tap-GEMM kernels share one epilogue (LF_TAPGEMM_EPILOGUE: bias, ReLU, masks, residual, BN sums); the whole-line kernel has
// its own (same arithmetic, output and operand tensors through LDS tiles).
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include <hip/hip_ext.h>

#include "lf_conv.h"
#include "lf_ldsdma.h"
#include "lf_types.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MT = 4;                 // 16-pixel tiles per wave
constexpr int WG_WAVES = 4;
constexpr int PIX_PER_WG = WG_WAVES * MT * 16;

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Buffer-addressed 16-byte load: SGPR resource + 32-bit VGPR byte offset + scalar byte offset.  Beside a dense MFMA stream a
// global_load whose 64-bit VGPR address was just computed costs the SIMD ~200 ns of matrix issue, this form ~6 ns
// (tools/vmem_cost.hip, profiles/r2_vmem_cost.txt); an offset >= num_records reads as 0.0f (the conv's zero padding).
typedef unsigned u32x4v __attribute__((vector_size(16)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ldb4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// (LF_OOB, the byte offset beyond every tensor the launcher admits, lives in lf_ldsdma.h)
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
// ReLU as a signed-integer maximum of the bit patterns (negative floats are negative integers): ONE v_max_i32 per element;
// fmaxf compiles to a canonicalising v_max_f32 v, v, v followed by the v_max_f32 with 0 -- twice the VALU work in the BN+ReLU
// operand prologue, where every VALU instruction issues beside the partner wave's MFMA stream
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 max0(f32x4 v) {
    i32x4 b = __builtin_bit_cast(i32x4, v);
    b.x = b.x > 0 ? b.x : 0; b.y = b.y > 0 ? b.y : 0; b.z = b.z > 0 ? b.z : 0; b.w = b.w > 0 ? b.w : 0;
    return __builtin_bit_cast(f32x4, b);
}
__device__ __forceinline__ f32x4 keep_pos(f32x4 v, f32x4 m) {
    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    return v;
}
// storage-typed access of the epilogue operands / result (S16: the tensors hold bf16 elements, see lf_types.h)
// (buffer-addressed like the operand loads: the epilogue of one wave runs beside its partner's MFMA stream, where 64-bit
// address arithmetic and VGPR-pair addressed memory instructions are what made it take 9 us instead of 4)
typedef unsigned u32x2v __attribute__((vector_size(8)));
template <bool S16>
__device__ __forceinline__ f32x4 epi_ld(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff = 0u) {   // element offsets
    if constexpr (S16) {
        const u32x2v q = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(off * 2u), (int)(soff * 2u), 0);
        f32x4 v;
        v.x = __uint_as_float(q[0] << 16); v.y = __uint_as_float(q[0] & 0xffff0000u);
        v.z = __uint_as_float(q[1] << 16); v.w = __uint_as_float(q[1] & 0xffff0000u);
        return v;
    } else return ldb4(r, off * 4u, soff * 4u);
}
template <bool S16>
__device__ __forceinline__ void epi_st(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    if constexpr (S16) {
        lf_bf16x4 b;
        b[0] = (lf_bf16)v.x; b[1] = (lf_bf16)v.y; b[2] = (lf_bf16)v.z; b[3] = (lf_bf16)v.w;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, b), r, (int)(off * 2u), 0, 0);
    } else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, (int)(off * 4u), 0, 0);   // (nt and sc0 sc1 hints measured: both -1.2 %)
}
__device__ __forceinline__ f32x4 round_bf16(f32x4 v) {
    v.x = (float)(lf_bf16)v.x; v.y = (float)(lf_bf16)v.y; v.z = (float)(lf_bf16)v.z; v.w = (float)(lf_bf16)v.w;
    return v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// one pair of fp32 values -> their packed bf16 roundings (v_cvt_pk_bf16_f32); v becomes the exact residuals
__device__ __forceinline__ unsigned split_pair(f32x2& v) {
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v.x -= __uint_as_float(u << 16);             // exact: the residual of a round-to-nearest fits fp32
    v.y -= __uint_as_float(u & 0xffff0000u);
    return u;
}
// sum over the 16 lanes that share l>>4 (one DPP row): lane pairs, quads (quad_perm), then the four quad totals by row
// rotations of 4 and 8 -- every lane ends with the row total (associated differently per quad; the caller reads lane 0 of the
// row).  DPP adds need neither the lane id nor the LDS crossbar that __shfl_xor (ds_bpermute) goes through.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// lane 0 of every 16-lane DPP row broadcast to its row (row_share:0 / row_newbcast:0)
__device__ __forceinline__ f32x4 bcast16(f32x4 v) {
    f32x4 r;
    r.x = dpp_mov<0x150>(v.x); r.y = dpp_mov<0x150>(v.y); r.z = dpp_mov<0x150>(v.z); r.w = dpp_mov<0x150>(v.w);
    return r;
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);     // row_ror:4
    v += dpp_mov<0x128>(v);     // row_ror:8
    return v;
}

#define LF_EPI_GROUPS 1      /* 4-wave groups per workgroup; the 512-thread split kernel redefines it */
#define LF_EPI_ONE_TILE false  /* tapgemm_kernel: true for the per-lane-row form of the three-tensor epilogue (one register short) */
#define LF_EPI_TID threadIdx.x  /* tapgemm_kernel: its per-tile laundered copy */
#define LF_EPI_ROW1 false      /* tapgemm_kernel: its ROW1 (a wave's 64 pixels in one image row) */
#define LF_EPI_PIVOT true      /* BN forward sums about a pivot (below); the run-time-flag forms of the 128-register split kernel and of the LDS-ring \
                                  kernels (cold paths: no network launch) have no registers for it and sum about 0 -- same row format */
#define LF_TAPGEMM_EPILOGUE \
    /* Pixel-tile outer, channel-tile inner: the loads of one operand tensor issued back to back cover one pixel's     \
     * contiguous channel run, so every cache line is touched once while it is hot (the channel-tile-outer order        \
     * revisited each line NT times with the whole grid's working set in between: 4x the HBM reads with bf16 tensors). \
     * The channel tiles of a pixel tile go in chunks of NC = 2 (one 128-byte line of fp32): with all NT tiles' operand \
     * loads in flight at once the three-tensor epilogues (ADD + MASK + BN-backward sums) needed 270+ registers and     \
     * spilled 13-29 of them to scratch.  NOBIAS: data-gradient epilogues (compiled-in flags) carry no bias vector. */  \
    const bool stats = (epi & (LF_EPI_STATS_SQ | LF_EPI_STATS_XHAT)) != 0; \
    constexpr bool NOBIAS = EPIC >= 0 && (EPIC & (LF_EPI_MASK | LF_EPI_ADD | LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) != 0; \
    constexpr int NC = (NT % 2 == 0 && !LF_EPI_ONE_TILE) ? 2 : 1; \
    /* MASKBN + STATS_XHAT: the mask's two per-channel vectors are re-read (L1) instead of held -- 32 registers, the difference \
     * between two and three waves per SIMD for that variant */ \
    constexpr bool HOISTM = HOISTV && !(EPIC >= 0 && (EPIC & LF_EPI_MASKBN) && (EPIC & LF_EPI_STATS_XHAT)); \
    __builtin_amdgcn_s_setprio(3);   /* ahead of the partner wave's MFMA stream: the sooner this wave retires, the sooner its slot refills */ \
    const __amdgpu_buffer_rsrc_t r_dst = make_rsrc(a.dst, 0xffffffffu), r_add = make_rsrc(a.add_src, 0xffffffffu), \
                                 r_msk = make_rsrc(a.mask_src, 0xffffffffu), r_aux = make_rsrc(a.aux, 0xffffffffu), \
                                 r_bias = make_rsrc(a.bias, 0xffffffffu), r_msc = make_rsrc(a.msc, 0xffffffffu), \
                                 r_msh = make_rsrc(a.msh, 0xffffffffu), r_dm = make_rsrc(a.dm, 0xffffffffu); \
    /* BatchNorm FORWARD statistics (STATS_SQ) are accumulated about a PIVOT: per channel the value of the wave's first pixel   \
     * (lane 0 of the 16-lane row, broadcast by DPP): sum (v - piv), sum (v - piv)^2.  With |v - piv| ~ sigma neither sum loses the  \
     * digits that sum v^2 loses when |mean| >> sigma -- E[v^2] - mean^2 from fp32 partials kept 2^-24 (mean / sigma)^2 of the    \
     * variance: 2.3e-4 on a parameter gradient at 50 sigma, where nn.BatchNorm2d has no such term (round 6).  A wave turns its   \
     * sums into (sum v, M2 = sum (v - mean_wave)^2) of its 64 pixels, the workgroup's four waves are merged with Chan's formula    \
     * (differences of wave means, never squares of sums), and the partial row of a 256-pixel tile is [sum v][M2 about the tile's  \
     * own mean]; the finalise kernel merges the tiles in fp64 (LfStatPart::tile_pix). */ \
    f32x4 s1[NT], s2[NT], bs[NOBIAS ? 1 : NT], hv[HOISTV ? NT : 1][2], piv[LF_EPI_PIVOT ? NT : 1]; \
_Pragma("unroll") \
    for (int n = 0; n < NT; ++n) { \
        const int co = cob + n * 16 + kq * 4; \
        s1[n] = zero4(); s2[n] = zero4(); piv[LF_EPI_PIVOT ? n : 0] = zero4(); \
        if constexpr (!NOBIAS) bs[n] = a.bias ? ldb4(r_bias, co * 4u, 0u) : zero4(); \
        if (HOISTV) { \
            if (HOISTM && (epi & LF_EPI_MASKBN)) { hv[n][0] = ldb4(r_msc, co * 4u, 0u); hv[n][1] = ldb4(r_msh, co * 4u, 0u); } \
        } \
    } \
_Pragma("unroll") \
    for (int m = 0; m < MT; ++m) { \
        const unsigned dbase = (unsigned)(((pn[m] * g.Hd + pi[m] * g.dsh + g.dah) * g.Wd + pj[m] * g.dsw + g.daw) * g.d_pix + g.d_choff + cob + kq * 4); \
_Pragma("unroll") \
        for (int n0 = 0; n0 < NT; n0 += NC) { \
        f32x4 la[NC], lm[NC], lx[NC], ld[NC]; \
_Pragma("unroll") \
        for (int j = 0; j < NC; ++j) { \
            const int n = n0 + j; \
            if (epi & LF_EPI_ADD) la[j] = epi_ld<S16>(r_add, dbase + n * 16); \
            if (epi & LF_EPI_MASK) lm[j] = epi_ld<S16>(r_msk, dbase + n * 16); \
            if (epi & (LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) lx[j] = epi_ld<S16>(r_aux, dbase + n * 16); \
            if ((epi & LF_EPI_STATS_XHAT) && a.dm && !LF_EPI_ROW1) ld[j] = ldb4(r_dm, (unsigned)(pn[m] * g.Cd + cob + n * 16 + kq * 4) * 4u, 0u); \
        } \
_Pragma("unroll") \
        for (int j = 0; j < NC; ++j) { \
            const int n = n0 + j; \
            f32x4 v = acc[n][m]; \
            if constexpr (!NOBIAS) v += bs[n]; \
            if (epi & LF_EPI_ADD) v += la[j]; \
            if (epi & LF_EPI_MASK) v = keep_pos(v, lm[j]); \
            const int co = cob + n * 16 + kq * 4;   /* per-channel vectors: L1-resident, re-read instead of held in registers */ \
            if (epi & LF_EPI_MASKBN) v = keep_pos(v, lx[j] * (HOISTM ? hv[HOISTV ? n : 0][0] : ldb4(r_msc, co * 4u, 0u)) + (HOISTM ? hv[HOISTV ? n : 0][1] : ldb4(r_msh, co * 4u, 0u))); \
            if (epi & LF_EPI_RELU) v = max0(v); \
            if (S16) v = round_bf16(v);   /* statistics are taken from the values as stored */ \
            if (pv[m]) epi_st<S16>(r_dst, dbase + n * 16, v); \
            if (!pv[m]) v = zero4(); \
            if (epi & LF_EPI_STATS_SQ) { \
                if (LF_EPI_PIVOT && m == 0) piv[LF_EPI_PIVOT ? n : 0] = bcast16(v); \
                f32x4 dv = LF_EPI_PIVOT ? v - piv[LF_EPI_PIVOT ? n : 0] : v; \
                if (!pv[m]) dv = zero4(); \
                s1[n] += dv; s2[n] += dv * dv; \
            } \
            if (epi & LF_EPI_STATS_XHAT) { \
                const f32x4 gm = (a.dm && !LF_EPI_ROW1) ? v * ld[j] : v; \
                s1[n] += gm; s2[n] += gm * lx[j];      /* RAW second sum: the finalise kernel applies x^ = t * rstd - mean * rstd in fp64 */ \
            } \
        } \
        } \
    } \
    if (LF_EPI_ROW1 && (epi & LF_EPI_STATS_XHAT) && a.dm) { \
        /* a wave's pixels lie in ONE image: the Dropout2d factor of (image, channel) scales the wave's sums once */ \
_Pragma("unroll") \
        for (int n = 0; n < NT; ++n) { \
            const f32x4 dmv = ldb4(r_dm, (unsigned)(pn[0] * g.Cd + cob + n * 16 + kq * 4) * 4u, 0u); \
            s1[n] *= dmv; s2[n] *= dmv; \
        } \
    } \
    if (stats) { \
        /* one partial row per 256-pixel tile: 4-wave group EG of the workgroup (EG = 0 in the 256-thread kernels) */ \
        const int EW = wave & 3, EG = wave >> 2; \
        __shared__ float sred[LF_EPI_GROUPS][WG_WAVES][NT][4][8]; \
_Pragma("unroll") \
        for (int n = 0; n < NT; ++n) { \
            f32x4 r1, r2; \
            r1.x = sum16(s1[n].x); r1.y = sum16(s1[n].y); r1.z = sum16(s1[n].z); r1.w = sum16(s1[n].w); \
            r2.x = sum16(s2[n].x); r2.y = sum16(s2[n].y); r2.z = sum16(s2[n].z); r2.w = sum16(s2[n].w); \
            if (epi & LF_EPI_STATS_SQ) {      /* pivot sums -> (sum v, M2 about the wave's own mean) of the wave's nw valid pixels */ \
                const unsigned lf_t0 = (bx * (unsigned)(WG_WAVES * LF_EPI_GROUPS) + (unsigned)wave) * 64u; \
                const float nw = lf_t0 < npix ? (float)min(npix - lf_t0, 64u) : 0.f; \
                const float rn = nw > 0.f ? 1.f / nw : 0.f; \
                r2 = r2 - r1 * r1 * rn; \
                r2.x = fmaxf(r2.x, 0.f); r2.y = fmaxf(r2.y, 0.f); r2.z = fmaxf(r2.z, 0.f); r2.w = fmaxf(r2.w, 0.f); \
                if (LF_EPI_PIVOT) r1 = r1 + piv[LF_EPI_PIVOT ? n : 0] * nw; \
            } \
            if (pl == 0) { \
                float* d = sred[EG][EW][n][kq]; \
                d[0] = r1.x; d[1] = r1.y; d[2] = r1.z; d[3] = r1.w; d[4] = r2.x; d[5] = r2.y; d[6] = r2.z; d[7] = r2.w; \
            } \
        } \
        __syncthreads(); \
        if ((LF_EPI_TID & 255) < NT * 4 * 8) { \
            const int tg = LF_EPI_TID >> 8, tt = LF_EPI_TID & 255; \
            const int j = tt & 7, q = (tt >> 3) & 3, n = tt >> 5; \
            float v = sred[tg][0][n][q][j] + sred[tg][1][n][q][j] + sred[tg][2][n][q][j] + sred[tg][3][n][q][j]; \
            if ((epi & LF_EPI_STATS_SQ) && j >= 4) {      /* Chan: M2 = sum_w M2_w + sum_w n_w (mean_w - mean)^2 */ \
                const unsigned lf_tb = (bx * (unsigned)LF_EPI_GROUPS + (unsigned)tg) * (unsigned)(WG_WAVES * 64); \
                float nwv[4], sw[4], nt = 0.f, st = 0.f; \
_Pragma("unroll") \
                for (int w = 0; w < 4; ++w) { \
                    const unsigned t0w = lf_tb + (unsigned)w * 64u; \
                    nwv[w] = t0w < npix ? (float)min(npix - t0w, 64u) : 0.f; \
                    sw[w] = sred[tg][w][n][q][j - 4]; \
                    nt += nwv[w]; st += sw[w]; \
                } \
                const float mg = nt > 0.f ? st / nt : 0.f; \
_Pragma("unroll") \
                for (int w = 0; w < 4; ++w) \
                    if (nwv[w] > 0.f) { const float dmw = sw[w] / nwv[w] - mg; v += nwv[w] * dmw * dmw; } \
            } \
            const int co = cob + n * 16 + q * 4 + (j & 3); \
            a.stats[((long)(j >> 2) * g.Cd + co) * a.stats_ld + ((long)bx * LF_EPI_GROUPS + tg)] = v;      /* channel-major rows */ \
        } \
    } \

extern __shared__ __attribute__((aligned(16))) unsigned char lf_tap_lds[];      // tap tables of tapgemm_kernel
constexpr size_t LF_TAP_LDS_PER_TAP = (size_t)WG_WAVES * 64 * (sizeof(uint4) + sizeof(unsigned));

// EPIC >= 0: the epilogue flags are compiled in (the combinations the network uses at 64 output channels per workgroup); the
// epilogue of one wave runs beside its partner's MFMA stream at ~14 cycles per VALU instruction, so the ~1500 instructions of
// the runtime-flag form (EPIC = -1) cost 9 us of a 30 us workgroup life -- and the slot it occupies cannot be refilled.
// (A one-operand-set form at <= 128 registers, four workgroups per CU so that the 4096 waves of a 64-channel launch run as ONE
// round, was measured and dropped: 72 vs 68 us -- three waves per SIMD of this form already overlap rounds.)
// ROW1 (Wl % 64 == 0): a wave's 64 pixels lie in one image row, so (image, row, first column) are wave-uniform and live in
// scalar registers: two divisions instead of eight in the prologue and ~10 vector registers less.
// DBG: per-wave phase stamps (tools/kbench.py --phases); compiled only into the instantiation lf_debug_conv1d_fwd_phases launches
// (the run-time-flag forms -- cold paths: the --clas trunk, kernel-level tests -- hold the union of all epilogue state beside the
// two accumulator sets and are given the whole register file, one wave per SIMD, instead of spilling 14-21 registers)
template <int NT, int PROC, int EPIC = -1, bool ROW1 = false, bool DBG = false>
__global__ __launch_bounds__(256, (EPIC >= 0 || NT < 4) ? 2 : 1) void tapgemm_kernel(const LfTapGeom g, const LfTapArgs a, const int pro, const int epi_rt) {
    const int epi = EPIC >= 0 ? EPIC : epi_rt;
    constexpr bool S16 = false, HOISTV = EPIC >= 0;  // compiled-in flags: the per-channel vectors are loaded once, after the loop
    unsigned long long tstamp[4] = {0ull, 0ull, 0ull, 0ull};
    if constexpr (DBG) tstamp[0] = __builtin_amdgcn_s_memrealtime();
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl);       // < 2^31, checked by the launcher
    // Workgroup b runs on XCD b % 8 (observed dispatch order).  Give every XCD a CONTIGUOUS range of pixel
    // tiles so that the halo rows neighbouring tiles share (and both halves of blockIdx.y) meet in one L2.
    // (A persistent loop over tiles and a 3-set operand ring were measured and dropped: at batch 32 every
    // layer is one or two rounds of resident workgroups, and the extra live state cost more than it hid.)
    // Round 3: the grid is sized to the workgroups the chip holds at once (launcher: occupancy x CUs) and a workgroup walks its
    // tiles first, first + stride, ... -- so WHICH workgroup (hence which CU: they are dealt breadth-first) processes the tiles
    // beyond the first round is fixed by construction, one extra tile per CU.  Left to the dispatcher, the 256 second-round
    // workgroups of a 64-channel launch (1024 workgroups, 768 resident) went to whichever CUs retired workgroups first: with the
    // three co-resident workgroups of a CU finishing together, a third of the CUs took three more and the rest none -- 71.5
    // instead of 58.5 us, decided by details as small as dead code behind the epilogue (r3 A/B runs, tools/ab_conv.py).
    const int cob = blockIdx.y * NT * 16;
    const unsigned ntiles = (npix + PIX_PER_WG - 1) / PIX_PER_WG;
    unsigned t_first = blockIdx.x, t_stride = gridDim.x, t_end = ntiles;
    if ((gridDim.x & 7u) == 0 && (ntiles & 7u) == 0) {
        const unsigned per = ntiles >> 3;
        t_first = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); t_stride = gridDim.x >> 3; t_end = (blockIdx.x & 7u) * per + per;
    }
    for (unsigned bx = t_first; bx < t_end; bx += t_stride) {
    // (the thread index is laundered per tile: hoisted out of the tile loop, the per-lane constants derived from it would
    // stay live across the whole body -- 10-20 registers, the difference between three and two waves per SIMD)
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, kq = lane >> 4;
    const unsigned tile0 = (bx * WG_WAVES + (ROW1 ? __builtin_amdgcn_readfirstlane(wave) : wave)) * (MT * 16);

    int pn[MT], pi[MT], pj[MT];
    bool pv[MT];
    if constexpr (ROW1) {
        const unsigned q = tile0 < npix ? tile0 : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        const int j0 = __builtin_amdgcn_readfirstlane((int)(q - r * (unsigned)g.Wl));
        const int n0 = __builtin_amdgcn_readfirstlane((int)(r / (unsigned)g.Hl));
        const int i0 = __builtin_amdgcn_readfirstlane((int)r) - n0 * g.Hl;
#pragma unroll
        for (int m = 0; m < MT; ++m) { pv[m] = tile0 < npix; pn[m] = n0; pi[m] = i0; pj[m] = j0 + m * 16 + pl; }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {   // 32-bit divisions: the 64-bit ones expand to ~100 instructions each
            const unsigned p = tile0 + m * 16 + pl;
            pv[m] = p < npix;
            const unsigned q = pv[m] ? p : 0u;
            const unsigned r = q / (unsigned)g.Wl;
            pj[m] = (int)(q - r * (unsigned)g.Wl);
            pn[m] = (int)(r / (unsigned)g.Hl);
            pi[m] = (int)(r - (unsigned)pn[m] * (unsigned)g.Hl);
        }
    }

    // Two accumulator sets (round 4): `acc` holds the running chain of ONE pair of K-steps (32 products per element), `accl`
    // the sum of the finished pairs.  One fp32 fma chain over the whole contraction (K = 192 at 64 channels, 384 at 128) left
    // the logits ~1.25x further from fp64 than oneDNN's fp32 convolution; tools/accum_study.py reproduces that on the CPU with
    // the kernel's summation order emulated (1.31x) and shows where it comes from (the 64-channel layers' K = 192 chain; the
    // folded BatchNorm and the fp32 statistics partials do not matter) and what removes it: 32-term segments summed into a
    // second accumulator -> 0.90x.  The flush is a VALU add per accumulator register and pair, issued between the MFMAs that
    // restart the chain from a zero C operand; the 64 extra registers put every variant at two waves per SIMD.
    f32x4 acc[NT][MT], accl[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[n][m] = zero4(); accl[n][m] = zero4(); }

    const int ncg = g.Cs >> 4;
    const int nsteps = g.ntaps * ncg;

    {
        // Branch-free main loop.  Per-tap source offsets and validity bits are computed ONCE
        // into an LDS table (one 16-byte row per lane and tap); the loop body is one basic block of two
        // K-steps (128 MFMAs + ~20 loads) with scalar selects for the (tap, channel-group) counters, so the
        // accumulators stay in place and the scheduler is free to sink loads between MFMAs.  Steps past the
        // end (odd step counts) are issued with clamped addresses and a zero mask.
        struct Step { f32x4 w[NT]; f32x4 x[MT]; f32x4 sc, sh; unsigned ok; };
        // (dynamic LDS, 5 KB per tap: the static 9-tap table was 46 KB and capped the CU at 3 workgroups)
        uint4 (*tab_off)[64] = reinterpret_cast<uint4 (*)[64]>(lf_tap_lds) + wave * g.ntaps;
        unsigned (*tab_ok)[64] = reinterpret_cast<unsigned (*)[64]>(lf_tap_lds + (size_t)WG_WAVES * g.ntaps * 64 * sizeof(uint4)) + wave * g.ntaps;
        // PROC == 0: a padding position holds the out-of-range offset LF_OOB, the buffer load returns the zero itself;
        // with the BN+ReLU prologue (transform(0) != 0) the offsets are clamped and the mask is applied after the transform.
        for (int t = 0; t < g.ntaps; ++t) {
            const int dh = g.tdh[t], dw = g.tdw[t];
            unsigned o[MT], okb = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int sy = pi[m] * g.ssh + dh, sx = pj[m] * g.ssw + dw;
                const bool in = pv[m] && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
                const int syc = min(max(sy, 0), g.Hs - 1), sxc = min(max(sx, 0), g.Ws - 1);
                o[m] = (unsigned)(((pn[m] * g.Hs + syc) * g.Ws + sxc) * g.s_pix + g.s_choff + kq * 4) * 4u;
                if (PROC == 0 && !in) o[m] = LF_OOB;
                okb |= (in ? 1u : 0u) << m;
            }
            tab_off[t][lane] = make_uint4(o[0], o[1], o[2], o[3]);
            tab_ok[t][lane] = okb;
        }
        // (each lane reads back only what it wrote: no barrier needed)
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.src, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 4, (long)LF_OOB));
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wp, 0xffffffffu), rsc = make_rsrc(a.pro_sc, 0xffffffffu), rsh = make_rsrc(a.pro_sh, 0xffffffffu);
        const unsigned wlane = (unsigned)(kq * g.Cd + cob + pl) * 16u;      // bytes
        const int wstep = g.Cd * 64;                       // bytes per 16-channel step
        const int ntaps = g.ntaps;
        const int wlast = (nsteps - 1) * wstep;
        int t_ld = 0, cg_ld = 0, wofs = 0;
        auto issue = [&](Step& S) {
            const bool live = t_ld < ntaps;
            const int tc = live ? t_ld : ntaps - 1;
            uint4 o = tab_off[tc][lane];
            const unsigned okb = tab_ok[tc][lane];
#pragma unroll
            for (int n = 0; n < NT; ++n) S.w[n] = ldb4(rw, wlane + n * 256, (unsigned)wofs);
            const unsigned c16 = (unsigned)cg_ld * 64u;    // bytes
            if constexpr (PROC == 0) {                     // a dead step (odd step count) reads zeros
                const unsigned dead = live ? 0u : LF_OOB;
                o.x |= dead; o.y |= dead; o.z |= dead; o.w |= dead;
            }
            S.x[0] = ldb4(rx, o.x, c16);
            S.x[1] = ldb4(rx, o.y, c16);
            S.x[2] = ldb4(rx, o.z, c16);
            S.x[3] = ldb4(rx, o.w, c16);
            if constexpr (PROC == LF_PRO_BNRELU) { S.sc = ldb4(rsc, (unsigned)kq * 16u, c16); S.sh = ldb4(rsh, (unsigned)kq * 16u, c16); }
            S.ok = live ? okb : 0u;
            // advance (scalar selects, no branches); a dead step re-reads the last live operands
            const int cgn = cg_ld + 1;
            const bool wrap = cgn == ncg;
            wofs = min(wofs + wstep, wlast);                // a dead step re-reads the last live weights
            cg_ld = live ? (wrap ? 0 : cgn) : cg_ld;
            t_ld = (live && wrap) ? t_ld + 1 : t_ld;
        };
        auto finish = [&](Step& S) {
            if constexpr (PROC == LF_PRO_BNRELU) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 v = max0(S.x[m] * S.sc + S.sh);
                    const bool in = (S.ok >> m) & 1u;
                    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
                    S.x[m] = v;
                }
            }
        };
        auto mma = [&](const Step& S, int s) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(S.w[n][s], S.x[m][s], acc[n][m], 0, 0, 0);
        };
        // first quarter of a pair: every accumulator is flushed into the long-term set and its chain restarted (C = 0).  The
        // add of tile (n, m) reads what the LAST quarter of the previous pair wrote 16 MFMAs ago -- no dependency stall -- and
        // issues in the shadow of the neighbouring MFMAs.
        // (64-product segments -- the even / odd tiles flushing in turn, two pairs per iteration, half the adds -- were measured:
        // +0.45 % on the step, logits 0.92x instead of 0.85x; the 32-product form stays: DESIGN.md section 9)
        auto mma_restart = [&](const Step& S) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    accl[n][m] += acc[n][m];
                    // (pinned: without the empty asm the IR passes sink all 64 adds behind the next loads, and the restarted
                    // chains then live in a THIRD register set beside the old one -- 256 registers, 13-65 of them spilled)
                    asm volatile("" : "+v"(accl[n][m]));
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(S.w[n][0], S.x[m][0], zero4(), 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // 2 VALU (v_pk_add_f32 x 2) ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // ... then this tile's MFMA
                }
        };
        static_assert(MT == 4, "tab_off packs 4 pixel tiles");
        if constexpr (DBG) tstamp[1] = __builtin_amdgcn_s_memrealtime();
        Step A, B;
        issue(A);
        finish(A);
        const int npairs = (nsteps + 1) >> 1;
        // (Priority falling with progress makes the two waves of a SIMD finish together instead of ~12 us apart -- measured
        // neutral at 128 channels and 15 % SLOWER at 64, where a third / fourth wave waits for the slot of the first finisher.)
        // Step order, pinned by scheduling fences: first quarter of this step's MFMAs, the next step's loads, two more quarters,
        // then the operand prologue of the next step (BN+ReLU / BatchNorm-backward transform: pure VALU work on the OTHER
        // register set, its loads had 32 MFMAs to land) interleaved with the last quarter -- the prologue's VALU instructions
        // issue in the shadow of MFMAs instead of in front of them.  (Where the loads sit inside the step does not matter at
        // three waves per SIMD: r3 sweep over six positions.)
        for (int pr = 0; pr < npairs; ++pr) {
            mma_restart(A);
            __builtin_amdgcn_sched_barrier(0);
            issue(B);                                   // (behind the first quarter: its eight dead operand registers are reused)
            __builtin_amdgcn_sched_barrier(0);
            mma(A, 1); mma(A, 2);
            __builtin_amdgcn_sched_barrier(0);
            finish(B);
            mma(A, 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(B, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue(A);
            __builtin_amdgcn_sched_barrier(0);
            mma(B, 1); mma(B, 2);
            __builtin_amdgcn_sched_barrier(0);
            finish(A);
            mma(B, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] += accl[n][m];
    if constexpr (!ROW1) {
        // per-lane pixel coordinates are RECOMPUTED for the epilogue (from a laundered thread index, so that the compiler cannot
        // keep the prologue's copies): 16 registers that would otherwise live across the main loop, where two accumulator sets
        // and two operand sets leave no room for them (24-32 registers spilled around the loop in the BatchNorm-backward-sum
        // variants).  The ROW1 form holds them in scalar registers.
        unsigned tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const unsigned tile0b = (bx * WG_WAVES + (tid2 >> 6)) * (MT * 16);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned p = tile0b + m * 16 + (tid2 & 15u);
            pv[m] = p < npix;
            const unsigned q = pv[m] ? p : 0u;
            const unsigned r = q / (unsigned)g.Wl;
            pj[m] = (int)(q - r * (unsigned)g.Wl);
            pn[m] = (int)(r / (unsigned)g.Hl);
            pi[m] = (int)(r - (unsigned)pn[m] * (unsigned)g.Hl);
        }
    }
    if constexpr (DBG) {   // make the stamp wait for the last MFMA: touch one accumulator
        asm volatile("" ::"v"(acc[0][0][0]));
        tstamp[2] = __builtin_amdgcn_s_memrealtime();
    }
#undef LF_EPI_ONE_TILE
#define LF_EPI_ONE_TILE (!ROW1 && EPIC == (LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT))
#undef LF_EPI_TID
#define LF_EPI_TID tid
#undef LF_EPI_ROW1
#define LF_EPI_ROW1 ROW1
    LF_TAPGEMM_EPILOGUE
#undef LF_EPI_ONE_TILE
#define LF_EPI_ONE_TILE false
#undef LF_EPI_TID
#define LF_EPI_TID threadIdx.x
#undef LF_EPI_ROW1
#define LF_EPI_ROW1 false
    if (stats && bx + t_stride < t_end) __syncthreads();      // the statistics staging area is reused by the next tile
    __builtin_amdgcn_s_setprio(0);
    }   // tiles of this workgroup
    if constexpr (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[3] = __builtin_amdgcn_s_memrealtime();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0 && a.dbg) {
            unsigned long long* d = a.dbg + ((unsigned long long)(blockIdx.y * gridDim.x + blockIdx.x) * WG_WAVES + wave) * 8;
            d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3];
            unsigned hwid, xcc;                   // which SIMD the wave ran on (tools/kbench.py --phases pairs the waves up)
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[4] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bf16 matrix-core kernel, operands streamed into registers (precision mode "bf16": bf16 tensors in HBM, fp32 accumulate; the
// launches the LDS kernels below do not take: operand prologue, ragged widths, tap tables with few pixels).
// v_mfma_f32_16x16x32_bf16: a lane supplies 8 consecutive k of its row / column (k-block kq = l>>4 of the 32-channel step), C/D
// layout as the fp32 form -- so the tap table, the pixel mapping and the whole epilogue are shared with tapgemm_kernel.  Per
// 32-channel step a lane loads ONE dwordx4 of its pixel (8 bf16 channels: the operand as it is), or, with the BN+ReLU prologue,
// widens it, applies the prologue and the validity mask in fp32 and converts back with v_cvt_pk_bf16_f32 (round to nearest even);
// 16 MFMAs (one per 16x16 tile) per step.  Weights are pre-packed bf16 [tap][ceil(Cs/32)*4][Cd][8], zero-padded to whole
// 32-channel steps, so the partial last step of a 16- or 48-channel contraction multiplies (valid, clamped) pixel data by zeros.
// At the bf16 rate (16 cycles per MFMA) the loop is bound by the operand path, not by the matrix cores.
// (Through round 5 the kernel also took fp32 tensors -- precision mode "bf16_mfma": removed, no BASELINE configuration used it.)
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 cvt_bf16x8(f32x4 lo, f32x4 hi) {
    bf16x8 r;
    r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
    r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
    return r;
}

// EPIC >= 0: epilogue flags compiled in (as in tapgemm_kernel).  FAST (no prologue, whole 32-channel steps): a padding position
// holds the out-of-range offset and reads as zero bits, the channel step is a scalar offset -- no per-step VALU work at all on
// bf16 tensors, where the 16 MFMAs of a step take 256 cycles and every VALU instruction beside them ~13.
template <int NT, int PROC, int EPIC = -1, bool FAST = false>
__global__ __launch_bounds__(256, 2) void tapgemm_bf16_kernel(const LfTapGeom g, const LfTapArgs a, const int pro, const int epi_rt) {
    static_assert(!FAST || PROC == 0, "FAST: the out-of-range zero must be the operand itself");
    const int epi = EPIC >= 0 ? EPIC : epi_rt;
    constexpr bool HOISTV = true, S16 = true;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, kq = lane >> 4;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl);
    const int cob = blockIdx.y * NT * 16;
    unsigned bx = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const unsigned tile0 = (bx * WG_WAVES + wave) * (MT * 16);

    int pn[MT], pi[MT], pj[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const unsigned p = tile0 + m * 16 + pl;
        pv[m] = p < npix;
        const unsigned q = pv[m] ? p : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj[m] = (int)(q - r * (unsigned)g.Wl);
        pn[m] = (int)(r / (unsigned)g.Hl);
        pi[m] = (int)(r - (unsigned)pn[m] * (unsigned)g.Hl);
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = zero4();

    // one dwordx4 per tile is the whole 8-channel operand (xl holds raw bf16 bits)
    struct Step { u32x4 w[NT]; f32x4 xl[MT]; f32x4 sc0, sc1, sh0, sh1; unsigned ok; };
    __shared__ uint4 tab_off[WG_WAVES][LF_MAX_TAPS][64];
    __shared__ unsigned tab_ok[WG_WAVES][LF_MAX_TAPS][64];
    for (int t = 0; t < g.ntaps; ++t) {
        const int dh = g.tdh[t], dw = g.tdw[t];
        unsigned o[MT], okb = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int sy = pi[m] * g.ssh + dh, sx = pj[m] * g.ssw + dw;
            const bool in = pv[m] && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
            const int syc = min(max(sy, 0), g.Hs - 1), sxc = min(max(sx, 0), g.Ws - 1);
            o[m] = (unsigned)(((pn[m] * g.Hs + syc) * g.Ws + sxc) * g.s_pix + g.s_choff);
            if constexpr (FAST) o[m] = in ? (o[m] + kq * 8) * 2u : LF_OOB;      // bytes, this lane's 8 channels
            okb |= (in ? 1u : 0u) << m;
        }
        tab_off[wave][t][lane] = make_uint4(o[0], o[1], o[2], o[3]);
        tab_ok[wave][t][lane] = okb;
    }
    const int ncb = (g.Cs + 31) >> 5;                       // 32-channel steps per tap
    const int nsteps = g.ntaps * ncb;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wp16, 0xffffffffu),
                                 rx = make_rsrc(a.src, FAST ? (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 2, (long)LF_OOB) : 0xffffffffu),
                                 rsc = make_rsrc(a.pro_sc, 0xffffffffu), rsh = make_rsrc(a.pro_sh, 0xffffffffu);
    const unsigned wlane = (unsigned)(kq * g.Cd + cob + pl) * 16u;      // bytes (8 bf16 per lane and tile)
    const int wstep = g.Cd * 32;                            // bf16 elements per 32-channel step
    const int ntaps = g.ntaps;
    const int wlast = (nsteps - 1) * wstep;
    int t_ld = 0, cb_ld = 0, wofs = 0;
    auto issue = [&](Step& S) {
        const bool live = t_ld < ntaps;
        const int tc = live ? t_ld : ntaps - 1;
        const uint4 o = tab_off[wave][tc][lane];
        const unsigned okb = tab_ok[wave][tc][lane];
#pragma unroll
        for (int n = 0; n < NT; ++n) S.w[n] = __builtin_bit_cast(u32x4, ldb4(rw, wlane + n * 256, (unsigned)wofs * 2u));
        const int c8 = min(cb_ld * 32 + kq * 8, g.Cs - 8);  // a partial last step re-reads valid channels (weights are 0)
        if constexpr (FAST) {
            const unsigned dead = live ? 0u : LF_OOB;        // a dead step (odd step count) reads zeros
            const unsigned cs = (unsigned)cb_ld * 64u;
            S.xl[0] = ldb4(rx, o.x | dead, cs); S.xl[1] = ldb4(rx, o.y | dead, cs);
            S.xl[2] = ldb4(rx, o.z | dead, cs); S.xl[3] = ldb4(rx, o.w | dead, cs);
        } else {                                             // buffer-addressed (see ldb4): bf16 elements, 16 bytes = 8 channels
            S.xl[0] = ldb4(rx, (o.x + c8) * 2u, 0u);
            S.xl[1] = ldb4(rx, (o.y + c8) * 2u, 0u);
            S.xl[2] = ldb4(rx, (o.z + c8) * 2u, 0u);
            S.xl[3] = ldb4(rx, (o.w + c8) * 2u, 0u);
        }
        if constexpr (PROC == LF_PRO_BNRELU) {
            S.sc0 = ldb4(rsc, c8 * 4u, 0u); S.sc1 = ldb4(rsc, c8 * 4u + 16u, 0u);
            S.sh0 = ldb4(rsh, c8 * 4u, 0u); S.sh1 = ldb4(rsh, c8 * 4u + 16u, 0u);
        }
        S.ok = live ? okb : 0u;
        const int cbn = cb_ld + 1;
        const bool wrap = cbn == ncb;
        wofs = min(wofs + wstep, wlast);                    // a dead step re-reads the last live weights
        cb_ld = live ? (wrap ? 0 : cbn) : cb_ld;
        t_ld = (live && wrap) ? t_ld + 1 : t_ld;
    };
    auto mma = [&](const Step& S) {
        bf16x8 xb[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool in = FAST || ((S.ok >> m) & 1u);
            if constexpr (PROC != LF_PRO_BNRELU) {       // raw bf16 bits: mask and use as they are
                u32x4 r = __builtin_bit_cast(u32x4, S.xl[m]);
                r.x = in ? r.x : 0u; r.y = in ? r.y : 0u; r.z = in ? r.z : 0u; r.w = in ? r.w : 0u;
                xb[m] = __builtin_bit_cast(bf16x8, r);
                continue;
            }
            f32x4 lo, hi;
            {
                const u32x4 r = __builtin_bit_cast(u32x4, S.xl[m]);
                lo.x = __uint_as_float(r.x << 16); lo.y = __uint_as_float(r.x & 0xffff0000u);
                lo.z = __uint_as_float(r.y << 16); lo.w = __uint_as_float(r.y & 0xffff0000u);
                hi.x = __uint_as_float(r.z << 16); hi.y = __uint_as_float(r.z & 0xffff0000u);
                hi.z = __uint_as_float(r.w << 16); hi.w = __uint_as_float(r.w & 0xffff0000u);
            }
            if constexpr (PROC == LF_PRO_BNRELU) { lo = max0(lo * S.sc0 + S.sh0); hi = max0(hi * S.sc1 + S.sh1); }
            lo.x = in ? lo.x : 0.f; lo.y = in ? lo.y : 0.f; lo.z = in ? lo.z : 0.f; lo.w = in ? lo.w : 0.f;
            hi.x = in ? hi.x : 0.f; hi.y = in ? hi.y : 0.f; hi.z = in ? hi.z : 0.f; hi.w = in ? hi.w : 0.f;
            xb[m] = cvt_bf16x8(lo, hi);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 wb = __builtin_bit_cast(bf16x8, S.w[n]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xb[m], acc[n][m], 0, 0, 0);
        }
    };
    static_assert(MT == 4, "tab_off packs 4 pixel tiles");
    Step A, B;
    issue(A);
    const int npairs = (nsteps + 1) >> 1;
    for (int pr = 0; pr < npairs; ++pr) {
        issue(B);
        mma(A);
        issue(A);
        mma(B);
    }
#undef LF_EPI_PIVOT
#define LF_EPI_PIVOT (EPIC >= 0 || NT < 4)      /* (the 64-channel-slab run-time-flag forms have no registers for the pivot: 282-810 spilled) */
    LF_TAPGEMM_EPILOGUE
#undef LF_EPI_PIVOT
#define LF_EPI_PIVOT true
}

// ---------------------------------------------------------------------------------------
// bf16 tensors through LDS (round 4): the launches of precision mode "bf16" without an operand prologue.  Two kernels,
// tapgemm_bf16_ring_kernel (any tap table; 32-channel K-steps) and tapgemm_bf16_wl_kernel (the 3-tap convolutions, whole
// 128-byte lines), share the machinery below: operands travel by `buffer_load_dwordx4 ... lds` (LDS-DMA: no register is held
// while the load is in flight) into a ring of LDS stages; the ring is ordered by hand --
//     s_waitcnt vmcnt(N) lgkmcnt(0)   my DMA of this step has landed (N instructions of younger steps may be in flight) and my
//                                     fragment reads of the step before have retired
//     s_barrier                       ... everyone's
//     issue the step that is R - 1 ahead into the stage the previous step occupied; read fragments; MFMAs
// -- with the bare s_barrier, because __syncthreads() carries a release fence that drains vmcnt to 0, i.e. the whole ring, every
// step.  The lgkmcnt(0) is not optional: hipcc moves a step's last MFMAs, and the waits for their operands, behind the next
// barrier, and a DMA instruction whose pixels are all padding returns its zeros without a memory round trip and overtakes those
// reads (wrong tiles in one launch of three at dilation 8 and row width 80 before the wait was there).
// ---------------------------------------------------------------------------------------
constexpr int LB_STAGES = 3;
constexpr int LB_X_BYTES = WG_WAVES * 64 * 64;          // 16 KB
constexpr int LB_W_BYTES = 4 * 64 * 16;                 // 4 KB
constexpr int LB_STAGE_BYTES = LB_X_BYTES + LB_W_BYTES;

// (lds_dma16, make_rsrc_words, wait_vm_lgkm0: lf_ldsdma.h)

// ---------------------------------------------------------------------------------------
// tapgemm_bf16_ring_kernel: LDS-staged bf16 tap-GEMM for ANY tap table (the 9-tap stride-2 convolution, the transposed-convolution
// phases, their gradients; the 3-tap convolutions go to tapgemm_bf16_wl_kernel), 64-output-channel slabs, 32-channel K-steps.
//
// Stage (20 KB) = X [wave 4][pixel 64][slot 4][16 B] + W [k-block 4][cout 64][16 B], 3 stages: two workgroups per CU.
//  * X: DMA instruction m of wave w carries pixels w*64 + m*16 + (lane >> 2), 16 bytes (8 channels) per lane, lane-linear in LDS
//    (that is what LDS-DMA does); a pixel's four 16-byte slots are XOR-swizzled by (pixel >> 2) & 3 -- applied to the SOURCE channel
//    block of the lane -- so that the fragment read of 8 consecutive lanes touches 8 different bank groups.  A padding tap
//    position carries the out-of-range offset: the DMA writes zeros.
//  * W: the packed bf16 weights [tap][k-block][Cd][8] make a (tap, 32-channel) step's 64-channel slab four 1 KB runs: one DMA
//    instruction per wave, fragment reads contiguous.
//  * PERSISTENT: a workgroup walks its work items (pixel tile x output-channel slab) and its load cursor runs two K-steps ahead
//    of the compute cursor ACROSS item boundaries: the first operands of the next item land while this item's epilogue stores
//    (the one-tile form of this kernel spent 2.7 of its 13 us building a per-lane tap table and 2.8 in the epilogue with the
//    memory pipe idle: profiles/r4_bf16_phases_one_tile.txt).
//  * No tap table: the launcher admits geometries whose 16-pixel groups lie in one image row (Wl % 16 == 0), so a DMA
//    instruction's 16 pixels share (image, row) -- wave-uniform, scalar registers: ONE division pair per wave and item -- and a
//    lane's byte offset is a scalar base plus a per-lane constant, with the column test (one unsigned compare) selecting the
//    out-of-range offset for padding.
// K order = tapgemm_bf16_kernel's: results bit-identical (tools/bf16_ab.py).  (A form owning all 128 output channels per
// workgroup -- X through the ring once per tap -- and the one-tile form are in DESIGN.md section 9.)
// ---------------------------------------------------------------------------------------
struct RingGroups { int n[4], i[4], j[4]; bool ok[4]; };
// the four consecutive 16-pixel groups from group index G0 on: (image, row, first column), all wave-uniform
__device__ __forceinline__ void groups_at(unsigned G0, int Hl, unsigned GR, unsigned ngroups, RingGroups& o) {
    const unsigned Gc = G0 < ngroups ? G0 : 0u;
    const unsigned R = Gc / GR;
    int jg = __builtin_amdgcn_readfirstlane((int)(Gc - R * GR));
    int n = __builtin_amdgcn_readfirstlane((int)(R / (unsigned)Hl));
    int i = __builtin_amdgcn_readfirstlane((int)R) - n * Hl;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        o.ok[m] = G0 + m < ngroups;
        o.n[m] = n; o.i[m] = i; o.j[m] = jg * 16;
        if (++jg == (int)GR) { jg = 0; if (++i == Hl) { i = 0; ++n; } }
    }
}
// ... of wave `wave` of the 256-pixel tile `tile`
__device__ __forceinline__ void ring_groups(unsigned tile, int wave, int Hl, unsigned GR, unsigned ngroups, RingGroups& o) {
    groups_at((tile * WG_WAVES + (unsigned)wave) * 4u, Hl, GR, ngroups, o);
}

// DBG: per-wave stamps, 16 x uint64 per wave: start, hardware id, then (K loop done, epilogue stores retired) per item, 7 items
template <int EPIC, bool DBG = false>
__global__ __launch_bounds__(256, 2) void tapgemm_bf16_ring_kernel(const LfTapGeom g, const LfTapArgs a, const int pro, const int epi_rt) {
    constexpr bool S16 = true, HOISTV = true;
    unsigned long long* dbgp = nullptr;
    int dbgk = 0;
    if constexpr (DBG) {
        dbgp = a.dbg + ((unsigned long long)blockIdx.x * WG_WAVES + (threadIdx.x >> 6)) * 16;
        if ((threadIdx.x & 63) == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            dbgp[0] = __builtin_amdgcn_s_memrealtime();
            dbgp[1] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
        }
    }
    constexpr int STAGE = LB_STAGE_BYTES;
    const int epi = EPIC >= 0 ? EPIC : epi_rt;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pl = lane & 15, kq = lane >> 4;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl), ngroups = npix >> 4, GR = (unsigned)g.Wl >> 4;
    const unsigned ntiles = (npix + PIX_PER_WG - 1) / PIX_PER_WG;
    const int slsh = g.Cd > 64 ? 1 : 0;                           // output-channel slabs per pixel tile: 1 or 2
    const unsigned nitems = ntiles << slsh;
    // work items first, first + stride, ... inside this XCD's contiguous range (see tapgemm_kernel)
    unsigned it_first = blockIdx.x, it_stride = gridDim.x, it_end = nitems;
    if ((gridDim.x & 7u) == 0 && (nitems & 7u) == 0) {
        const unsigned per = nitems >> 3;
        it_first = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); it_stride = gridDim.x >> 3; it_end = (blockIdx.x & 7u) * per + per;
    }
    // tap offsets: tap t in lane t (read back with v_readlane: no dependent scalar loads in the issue path)
    int tapv = 0;
#pragma unroll
    for (int t = 0; t < LF_MAX_TAPS; ++t)
        if (lane == t) tapv = (g.tdh[t] & 0xffff) | (g.tdw[t] << 16);

    const int ncb = g.Cs >> 5;
    const int nsteps = g.ntaps * ncb;
    const i32x4s rx = make_rsrc_words(a.src, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 2, (long)LF_OOB)),
                 rw = make_rsrc_words(a.wp16, 0xffffffffu);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lf_tap_lds;
    unsigned char* const stages = lf_tap_lds;
    const int wstep = g.Cd * 64;                               // bytes per 32-channel step (4 k-blocks x Cd x 16)
    const int spix2 = g.s_pix * 2;
    // per-lane constants of the DMA mapping: instruction m, this lane -> pixel (group m) + (lane >> 2), channel block
    // (lane & 3) ^ ((lane >> 4) & 3) of the 32-channel step
    const int dq = lane >> 2, dkq = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned lane_x = (unsigned)((dq * g.ssw * g.s_pix + g.s_choff + dkq * 8) * 2);
    const int lane_dx = dq * g.ssw;
    const unsigned lane_w = (unsigned)((wave * g.Cd + lane) * 16);

    // ---- load cursor (wave-uniform): item, tap, channel block, ring stage; the item's four groups as (pixel index of the
    // row position without the tap, row, column) and a validity mask
    unsigned it_ld = it_first;
    int t_ld = 0, cb_ld = 0, st_ld = 0, wofs = 0, cob_ld = 0;
    int lrow[4], liy[4], lsx[4];
    unsigned lok = 0;
    auto cursor_item = [&]() {
        lok = 0;
        if (it_ld < it_end) {
            RingGroups G;
            ring_groups(it_ld >> slsh, wave, g.Hl, GR, ngroups, G);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                liy[m] = G.i[m] * g.ssh; lsx[m] = G.j[m] * g.ssw;
                lrow[m] = (G.n[m] * g.Hs + liy[m]) * g.Ws + lsx[m];
                lok |= G.ok[m] ? 1u << m : 0u;
            }
            cob_ld = (int)(it_ld & (unsigned)slsh) * 64;
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) { liy[m] = 0; lsx[m] = 0; lrow[m] = 0; }
        }
    };
    cursor_item();
    auto issue = [&]() {                                     // DMA of the cursor's step into stage st_ld
        const unsigned st = lds0 + (unsigned)(st_ld * STAGE);
        const int tv = __builtin_amdgcn_readlane(tapv, t_ld);
        const int dh = (int)(short)(tv & 0xffff), dw = tv >> 16;
        const int tapoff = dh * g.Ws + dw;
        const unsigned xs = st + (unsigned)wave * 4096u;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int sy = liy[m] + dh;
            const bool yok = ((lok >> m) & 1u) && sy >= 0 && sy < g.Hs;
            const unsigned base = (unsigned)((lrow[m] + tapoff) * spix2 + cb_ld * 64);
            const bool in = yok && (unsigned)(lsx[m] + dw + lane_dx) < (unsigned)g.Ws;
            lds_dma16(rx, xs + (unsigned)m * 1024u, in ? base + lane_x : LF_OOB, 0u);
        }
        lds_dma16(rw, st + (unsigned)(LB_X_BYTES + wave * 1024), lane_w + (unsigned)(cob_ld * 16), (unsigned)wofs);
        // advance; past the last item the X part writes zeros (out-of-range offsets) into a stage nobody reads
        st_ld = st_ld == LB_STAGES - 1 ? 0 : st_ld + 1;
        wofs += wstep;
        if (++cb_ld == ncb) {
            cb_ld = 0;
            if (++t_ld == g.ntaps) { t_ld = 0; wofs = 0; it_ld += it_stride; cursor_item(); }
        }
    };
    issue();
    issue();
    // fragment addresses (bytes inside a stage)
    const unsigned xfrag = (unsigned)(wave * 4096 + pl * 64 + ((kq ^ (pl >> 2)) & 3) * 16);       // + m * 1024
    const unsigned wfrag = (unsigned)(LB_X_BYTES + (kq * 64 + pl) * 16);                           // + n * 256
    int st_c = 0;
    for (unsigned it = it_first; it < it_end; it += it_stride) {
        constexpr int NT = 4;
        f32x4 acc[NT][MT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = zero4();
        for (int s = 0; s < nsteps; ++s) {
            // my DMA of this step has landed (the instructions of the next step may still be in flight) ...
            asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // ... and everyone's; everyone is past the fragment reads of the step before
            asm volatile("" ::: "memory");
            issue();                             // two steps ahead -> the stage the previous step occupied
            const unsigned char* st = stages + st_c * STAGE;
            st_c = st_c == LB_STAGES - 1 ? 0 : st_c + 1;
            bf16x8 xb[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) xb[m] = *reinterpret_cast<const bf16x8*>(st + xfrag + m * 1024);
            bf16x8 wb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) wb[n] = *reinterpret_cast<const bf16x8*>(st + wfrag + n * 256);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[n], xb[m], acc[n][m], 0, 0, 0);
        }
        if constexpr (DBG) {
            asm volatile("" ::"v"(acc[0][0][0]));
            if (lane == 0 && dbgk < 7) dbgp[2 + 2 * dbgk] = __builtin_amdgcn_s_memrealtime();
        }
        // ---- epilogue of this item (the accumulator layout of tapgemm_bf16_kernel: tile m = group m, pixel m*16 + pl)
        const unsigned bx = it >> slsh;
        int pn[MT], pi[MT], pj[MT];
        bool pv[MT];
        {
            RingGroups G;
            ring_groups(bx, wave, g.Hl, GR, ngroups, G);
#pragma unroll
            for (int m = 0; m < MT; ++m) { pn[m] = G.n[m]; pi[m] = G.i[m]; pj[m] = G.j[m] + pl; pv[m] = G.ok[m]; }
        }
        const int cob = (int)(it & (unsigned)slsh) * 64;
#undef LF_EPI_PIVOT
#define LF_EPI_PIVOT (EPIC >= 0)
        LF_TAPGEMM_EPILOGUE
#undef LF_EPI_PIVOT
#define LF_EPI_PIVOT true
        __builtin_amdgcn_s_setprio(0);
        if constexpr (DBG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && dbgk < 7) dbgp[3 + 2 * dbgk] = __builtin_amdgcn_s_memrealtime();
            ++dbgk;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the two trailing (dead) steps: nothing may land in LDS after this
    __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------
// tapgemm_bf16_wl_kernel ("whole lines"): the 3-tap bf16 convolutions of non_bottleneck_1d (Cs = Cd = 64 or 128, stride 1).
//
// tools/l2_stream.hip measures what bounds every bf16 kernel above (profiles/r4_l2_stream.txt): an instruction whose 64 lanes
// touch 16 cache lines moves 16 lines' worth of L1 time whatever it uses of them.  The MFMA operand layout -- a lane = 16 bytes of
// pixel l & 15 -- makes a register load take 64 bytes of each of 16 lines (3 taps of a 52 MB tensor: 26.9 us, 5.9 TB/s; the same
// bytes as whole lines: 12.5 us, 12.5 TB/s), and the accumulator layout makes the epilogue store 8 bytes per lane, 32 of each
// line (+17 us for 52 MB against +7 us as whole lines).  All of the above kernels sit at that ~44 us; this one touches memory
// in whole 128-byte lines only:
//  * X: LDS-DMA instructions of 8 pixels x 128 bytes (lane -> pixel l >> 3, 16-byte chunk l & 7) into a ring of K-steps of
//    64 pixels x 64 channels (8 KB); the waves read MFMA fragments from it (chunk XOR (pixel >> 1) & 7 on the SOURCE side
//    makes the ds_read_b128 lane groups conflict-free);
//  * output: the epilogue writes bf16 quads into an LDS tile (16-byte chunks XOR pixel), barrier, then every wave stores
//    1 KB instructions of 4 pixels x 256 bytes (8 x 128 at 64 channels);
//  * W: NOT in LDS and not in the ring -- the workgroup's 64 pixels are shared by its four waves, each owning a quarter of the
//    output channels (wave tile 64 pixels x Cd/4), so a wave's weights are 3 taps x Cs x Cd/4 = 24 KB at 128 channels: 96
//    REGISTERS per lane, loaded once per persistent workgroup, statically indexed by the fully unrolled K loop.  The LDS
//    is ring (4 to 8 stages of 8 KB, by variant) + output tile + the epilogue's operand tiles (WlCfg below) = at most 80 KB: two
//    workgroups per CU, each with its ring's depth - 1 K-steps in flight.
// Work item = 256 pixels (the BN-statistics row of the other kernels) as four 64-pixel sub-tiles; the ring runs across
// sub-tiles and items.  K order = tapgemm_bf16_kernel's (tap, then channel): results bit-identical.
// ---------------------------------------------------------------------------------------
constexpr int WL_STAGE = 64 * 128;
// Epilogue operand tensors (the data gradient's ReLU-mask source, residual gradient, BN-backward operand) come in whole lines too: by
// LDS-DMA into tiles of the output tile's layout, issued behind the first K-step of the sub-tile they belong to and read by the
// epilogue in the accumulator layout (8 bytes per lane from LDS instead of 32 of every 128-byte line from L2: +40 us per tensor and
// launch at 80 x 160 x 64 images before).  The residual lands IN the output tile (each lane overwrites what it read); the other
// two have their own tiles, which come out of the ring's depth: 2 x 80 KB per CU.
template <int CB, int EPIC, int PROC = 0> struct WlCfg {
    static constexpr int NT = CB / 2, KS = CB / 2, NSTEP = 3 * KS, CD = CB * 32, PIXB = CD * 2;
    static constexpr int OUT_BYTES = 64 * PIXB;
    static constexpr bool ST_ADD = EPIC >= 0 && (EPIC & LF_EPI_ADD) != 0, ST_MSK = EPIC >= 0 && (EPIC & LF_EPI_MASK) != 0,
                          ST_AUX = EPIC >= 0 && (EPIC & (LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) != 0 &&
                                   !(CB == 4 && (EPIC & LF_EPI_MASKBN) != 0);          // (that variant has no registers left for it: 7 spilled)
    static constexpr int NBUF = (ST_MSK ? 1 : 0) + (ST_AUX ? 1 : 0);                  // tiles beside the output tile
    static constexpr int NI = OUT_BYTES / 1024 / WG_WAVES;                            // 1 KB instructions per wave and tile
    static constexpr int NAUXI = ((ST_ADD ? 1 : 0) + NBUF) * NI;                      // staging instructions per wave and sub-tile
    static constexpr int MAXST = CB == 4 ? 6 : 8;
    static constexpr int FIT = (80 * 1024 - OUT_BYTES * (1 + NBUF)) / WL_STAGE;
    static constexpr int STAGES = FIT < MAXST ? FIT : MAXST;
    static constexpr int TILES_END = STAGES * WL_STAGE + OUT_BYTES * (1 + NBUF);        // the prologue's scale / shift vectors live here
    static constexpr size_t LDS = (size_t)TILES_END + (PROC ? 2 * CD * 4 : 0);
    static_assert(STAGES >= 3, "ring too shallow");
};

// PROC = 1 (the block's third convolution: its operand is relu(bn1(t2)), never stored): behind its own vmcnt wait and in front of
// the step's barrier every wave transforms the 16 pixels x 64 channels IT brought in, in place in the stage (fp32 math on the
// widened values, tapgemm_bf16_kernel's expressions: results bit-identical), so the four waves that read the fragments do not
// each redo it; padding pixels (zeros from the DMA) are kept zero by the same column / row test that issued them.
template <int CB, int EPIC, int PROC = 0>
__global__ __launch_bounds__(256, 2) void tapgemm_bf16_wl_kernel(const LfTapGeom g, const LfTapArgs a, const int pro, const int epi_rt) {
    typedef WlCfg<CB, EPIC, PROC> C;
    constexpr int NT = C::NT, KS = C::KS, NSTEP = C::NSTEP, S = C::STAGES, CD = C::CD, PIXB = C::PIXB, NCH = PIXB / 16;
    constexpr bool S16 = true;
    const int epi = EPIC >= 0 ? EPIC : epi_rt;
    constexpr bool NOBIAS = EPIC >= 0 && (EPIC & (LF_EPI_MASK | LF_EPI_ADD | LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) != 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pl = lane & 15, kq = lane >> 4;
    const int cobw = wave * 16 * NT;                        // this wave's output channels
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl), ngroups = npix >> 4, GR = (unsigned)g.Wl >> 4;
    const unsigned nitems = (npix + PIX_PER_WG - 1) / PIX_PER_WG;
    unsigned it_first = blockIdx.x, it_stride = gridDim.x, it_end = nitems;
    if ((gridDim.x & 7u) == 0 && (nitems & 7u) == 0) {
        const unsigned per = nitems >> 3;
        it_first = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); it_stride = gridDim.x >> 3; it_end = (blockIdx.x & 7u) * per + per;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lf_tap_lds;
    unsigned char* const ring = lf_tap_lds;
    unsigned char* const otile = lf_tap_lds + S * WL_STAGE;
    unsigned char* const mtile = otile + C::OUT_BYTES;                                 // mask source (when staged)
    unsigned char* const xtile = otile + C::OUT_BYTES * (C::ST_MSK ? 2 : 1);          // BN-backward operand (when staged)

    // ---- this wave's weights -> registers: [tap][32-channel k-block][16-channel output tile], the MFMA A operand
    bf16x8 wr[3][CB][NT];
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wp16, 0xffffffffu);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int kb = 0; kb < CB; ++kb)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    wr[t][kb][n] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rw, (int)((((t * CB + kb) * 4 + kq) * CD + cobw + n * 16 + pl) * 16), 0, 0));
    }
    // tap t's offsets in lane t, read back with v_readlane (a select chain over three values indexed by the run-time tap becomes an
    // indexed array in scratch memory -- and the kernel-argument struct with it)
    int tapv = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (lane == t) tapv = (g.tdh[t] & 0xffff) | (g.tdw[t] << 16);
    const i32x4s rx = make_rsrc_words(a.src, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 2, (long)LF_OOB));
    const int spix2 = g.s_pix * 2;
    // DMA mapping: this wave carries group `wave` of the sub-tile (pixels wave*16 ..+15) as two instructions jj of 8 pixels;
    // lane -> pixel jj*8 + (lane >> 3), LDS slot lane & 7 = chunk XOR ((pixel >> 1) & 7)
    const int dpx = lane >> 3;
    unsigned lane_x[2];
    int lane_dx[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int chunk = (lane & 7) ^ ((jj * 4 + (lane >> 4)) & 7);
        lane_x[jj] = (unsigned)(((jj * 8 + dpx) * g.s_pix + g.s_choff) * 2 + chunk * 16);
        lane_dx[jj] = jj * 8 + dpx;
    }
    // ---- load cursor (wave-uniform): item, sub-tile, step; this wave's group of the cursor's sub-tile
    unsigned it_ld = it_first;
    int sub_ld = 0, s_ld = 0, st_ld = 0;
    int lrow = 0, liy = 0, lsx = 0;
    bool lok = false;
    auto cursor_sub = [&]() __attribute__((always_inline)) {
        lok = false;
        if (it_ld < it_end) {
            const unsigned G = (it_ld * 4u + (unsigned)sub_ld) * 4u + (unsigned)wave;
            lok = G < ngroups;
            const unsigned Gc = lok ? G : 0u;
            const unsigned R = Gc / GR;
            const int jg = __builtin_amdgcn_readfirstlane((int)(Gc - R * GR));
            const int n = __builtin_amdgcn_readfirstlane((int)(R / (unsigned)g.Hl));
            const int i = __builtin_amdgcn_readfirstlane((int)R) - n * g.Hl;
            liy = i; lsx = jg * 16;
            lrow = (n * g.Hs + i) * g.Ws + lsx;
        }
    };
    cursor_sub();
    auto issue = [&]() __attribute__((always_inline)) {
        const int t = KS == 1 ? s_ld : s_ld >> 1, ks = KS == 1 ? 0 : s_ld & 1;
        const int tv = __builtin_amdgcn_readlane(tapv, t);
        const int dh = (int)(short)(tv & 0xffff), dw = tv >> 16;
        const int sy = liy + dh;
        const bool yok = lok && sy >= 0 && sy < g.Hs;
        const unsigned base = (unsigned)((lrow + dh * g.Ws + dw) * spix2 + ks * 128);
        const unsigned dst = lds0 + (unsigned)(st_ld * WL_STAGE + wave * 2048);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const bool in = yok && (unsigned)(lsx + dw + lane_dx[jj]) < (unsigned)g.Ws;
            lds_dma16(rx, dst + (unsigned)jj * 1024u, in ? base + lane_x[jj] : LF_OOB, 0u);
        }
        st_ld = st_ld == S - 1 ? 0 : st_ld + 1;
        if (++s_ld == NSTEP) {
            s_ld = 0;
            if (++sub_ld == 4) { sub_ld = 0; it_ld += it_stride; }
            cursor_sub();
        }
    };
#pragma unroll
    for (int i = 0; i < S - 1; ++i) issue();

    float* const ptab = reinterpret_cast<float*>(lf_tap_lds + C::TILES_END);           // [2][CD]: scale, shift
    if constexpr (PROC == LF_PRO_BNRELU) {
        for (int c = threadIdx.x; c < CD; c += 256) { ptab[c] = a.pro_sc[c]; ptab[CD + c] = a.pro_sh[c]; }
        __syncthreads();
    }
    // ---- staging of the epilogue's operand tensors: wave w carries group w of the sub-tile (NI instructions of 1 KB per tensor);
    // instruction ii = w * NI + k is bytes ii*1024 .. +1023 of the pixel-major tile: lane -> pixel p, 16-byte slot = chunk XOR p
    // (resources bounded by the tensor: the out-of-range offset of a group beyond it reads zeros)
    const unsigned dbytes = (unsigned)min((long)g.N * g.Hd * g.Wd * g.d_pix * 2, (long)LF_OOB);
    const i32x4s ra_add = make_rsrc_words(a.add_src, dbytes), ra_msk = make_rsrc_words(a.mask_src, dbytes), ra_aux = make_rsrc_words(a.aux, dbytes);
    auto stage_tensors = [&](const RingGroups& G) __attribute__((always_inline)) {
        if constexpr (C::NAUXI > 0) {
            const int gn = wave == 0 ? G.n[0] : wave == 1 ? G.n[1] : wave == 2 ? G.n[2] : G.n[3];
            const int gi = wave == 0 ? G.i[0] : wave == 1 ? G.i[1] : wave == 2 ? G.i[2] : G.i[3];
            const int gj = wave == 0 ? G.j[0] : wave == 1 ? G.j[1] : wave == 2 ? G.j[2] : G.j[3];
            const bool gok = wave == 0 ? G.ok[0] : wave == 1 ? G.ok[1] : wave == 2 ? G.ok[2] : G.ok[3];
            const unsigned base = (unsigned)(((gn * g.Hd + gi) * g.Wd + gj) * g.d_pix * 2);
            const unsigned dst0 = lds0 + (unsigned)(S * WL_STAGE + wave * C::NI * 1024);
#pragma unroll
            for (int k = 0; k < C::NI; ++k) {
                const int ii = wave * C::NI + k;
                const int p = (ii * 1024) / PIXB + (lane * 16) / PIXB, slot = ((lane * 16) % PIXB) / 16;
                const unsigned st_lane = (unsigned)((((p & 15) * g.d_pix + g.d_choff) * 2) + ((slot ^ p) & (NCH - 1)) * 16);
                const unsigned vo = gok ? base + st_lane : LF_OOB;                // (a whole group beyond the tensor: zeros)
                if constexpr (C::ST_ADD) lds_dma16(ra_add, dst0 + (unsigned)(k * 1024), vo, 0u);
                if constexpr (C::ST_MSK) lds_dma16(ra_msk, dst0 + (unsigned)(C::OUT_BYTES + k * 1024), vo, 0u);
                if constexpr (C::ST_AUX) lds_dma16(ra_aux, dst0 + (unsigned)(C::OUT_BYTES * (C::ST_MSK ? 2 : 1) + k * 1024), vo, 0u);
            }
        }
    };

    // epilogue constants
    const __amdgpu_buffer_rsrc_t r_dst = make_rsrc(a.dst, 0xffffffffu), r_add = make_rsrc(a.add_src, 0xffffffffu),
                                 r_msk = make_rsrc(a.mask_src, 0xffffffffu), r_aux = make_rsrc(a.aux, 0xffffffffu),
                                 r_bias = make_rsrc(a.bias, 0xffffffffu), r_msc = make_rsrc(a.msc, 0xffffffffu),
                                 r_msh = make_rsrc(a.msh, 0xffffffffu), r_dm = make_rsrc(a.dm, 0xffffffffu);
    // fragment read addresses inside a stage: pixel m*16 + pl (128 B rows), chunk kb*4 + kq XOR (pl >> 1) & 7
    const unsigned xf0 = (unsigned)(pl * 128 + (((0 + kq) ^ (pl >> 1)) & 7) * 16), xf1 = (unsigned)(pl * 128 + (((4 + kq) ^ (pl >> 1)) & 7) * 16);
    const bool stats = (epi & (LF_EPI_STATS_SQ | LF_EPI_STATS_XHAT)) != 0;
    int st_c = 0;
    for (unsigned bx = it_first; bx < it_end; bx += it_stride) {
        constexpr bool PIVOT = EPIC >= 0;    // the pivot of the BN forward sums (see LF_TAPGEMM_EPILOGUE); the run-time-flag form sums about 0
        f32x4 s1[NT], s2[NT], piv[PIVOT ? NT : 1];
#pragma unroll
        for (int n = 0; n < NT; ++n) { s1[n] = zero4(); s2[n] = zero4(); piv[PIVOT ? n : 0] = zero4(); }
        for (int sub = 0; sub < 4; ++sub) {
            f32x4 acc[NT][MT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = zero4();
            RingGroups G;
            groups_at((bx * 4u + (unsigned)sub) * 4u, g.Hl, GR, ngroups, G);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                // my DMA of this step has landed: exactly S - 2 younger ring steps of 2 instructions may be in flight -- plus, behind
                // step 0, this sub-tile's staging instructions while the step's own DMA is older than they are (steps 1 .. S-1; the
                // count must not EXCEED the younger instructions, or the wait passes early) ...
                // ... and MY fragment reads of the step before have retired (lgkmcnt): hipcc moves that step's last MFMAs -- and the
                // waits for their operands -- behind the barrier, and the stage is restaged right behind it: a DMA instruction whose 8
                // pixels are all padding returns its zeros without a memory round trip and overtook those reads (wrong tiles in 1 of
                // 3 launches at dilation 8, row width 80)
                if (s >= 1 && s <= S - 1) wait_vm_lgkm0<2 * (S - 2) + C::NAUXI>();
                else wait_vm_lgkm0<2 * (S - 2)>();
                const int t = KS == 1 ? s : s >> 1, ks = KS == 1 ? 0 : s & 1;          // (compile-time after unrolling: wr is indexed statically)
                if constexpr (PROC == LF_PRO_BNRELU) {
                    // my part of this step's stage has landed: relu(bn(x)) on it, in place (the other waves read it behind the barrier)
                    unsigned char* mine = ring + st_c * WL_STAGE + wave * 2048;
                    const int tv = __builtin_amdgcn_readlane(tapv, t);
                    const int dh = (int)(short)(tv & 0xffff), dw = tv >> 16;
                    const int gi = wave == 0 ? G.i[0] : wave == 1 ? G.i[1] : wave == 2 ? G.i[2] : G.i[3];
                    const int gj = wave == 0 ? G.j[0] : wave == 1 ? G.j[1] : wave == 2 ? G.j[2] : G.j[3];
                    const bool gok = wave == 0 ? G.ok[0] : wave == 1 ? G.ok[1] : wave == 2 ? G.ok[2] : G.ok[3];
                    const int sy = gi + dh;
                    const bool yok = gok && sy >= 0 && sy < g.Hs;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const bool in = yok && (unsigned)(gj + dw + lane_dx[jj]) < (unsigned)g.Ws;
                        const int c0 = ks * 64 + (((lane & 7) ^ ((jj * 4 + (lane >> 4)) & 7)) * 8);      // this lane's 8 channels
                        u32x4v r = *reinterpret_cast<const u32x4v*>(mine + jj * 1024 + lane * 16);
                        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(ptab + c0), sc1 = *reinterpret_cast<const f32x4*>(ptab + c0 + 4),
                                    sh0 = *reinterpret_cast<const f32x4*>(ptab + CD + c0), sh1 = *reinterpret_cast<const f32x4*>(ptab + CD + c0 + 4);
                        f32x4 lo, hv;
                        lo.x = __uint_as_float(r[0] << 16); lo.y = __uint_as_float(r[0] & 0xffff0000u);
                        lo.z = __uint_as_float(r[1] << 16); lo.w = __uint_as_float(r[1] & 0xffff0000u);
                        hv.x = __uint_as_float(r[2] << 16); hv.y = __uint_as_float(r[2] & 0xffff0000u);
                        hv.z = __uint_as_float(r[3] << 16); hv.w = __uint_as_float(r[3] & 0xffff0000u);
                        lo = max0(lo * sc0 + sh0); hv = max0(hv * sc1 + sh1);
                        lo.x = in ? lo.x : 0.f; lo.y = in ? lo.y : 0.f; lo.z = in ? lo.z : 0.f; lo.w = in ? lo.w : 0.f;
                        hv.x = in ? hv.x : 0.f; hv.y = in ? hv.y : 0.f; hv.z = in ? hv.z : 0.f; hv.w = in ? hv.w : 0.f;
                        *reinterpret_cast<bf16x8*>(mine + jj * 1024 + lane * 16) = cvt_bf16x8(lo, hv);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my writes are in the stage
                }
                __builtin_amdgcn_s_barrier();        // ... and everyone's; everyone is past the fragment reads of the step before
                asm volatile("" ::: "memory");
                issue();                             // S - 1 steps ahead -> the stage the previous step occupied
                if (s == 0) stage_tensors(G);        // (everyone is past the previous sub-tile's reads of the tiles: the barrier above)
                const unsigned char* st = ring + st_c * WL_STAGE;
                st_c = st_c == S - 1 ? 0 : st_c + 1;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8 xb[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) xb[m] = *reinterpret_cast<const bf16x8*>(st + (kb ? xf1 : xf0) + m * 2048);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t][ks * 2 + kb][n], xb[m], acc[n][m], 0, 0, 0);
                }
            }
            // ---- epilogue of the sub-tile: accumulator tile (n, m) = channels cobw + n*16 + kq*4 .. +3 of pixel m*16 + pl
            if constexpr (C::NAUXI > 0) {            // the staged tensors have landed: mine (the ring steps issued since are younger) ...
                wait_vm_lgkm0<2 * (NSTEP - 1)>();
                __builtin_amdgcn_s_barrier();        // ... and everyone's
                asm volatile("" ::: "memory");
            }
            // (128 channels, MASKBN + BN-backward sums: the mask's per-channel vectors are re-read from L1 per use -- held, they are the
            // 9 registers that variant would spill beside its 96 weight registers)
            constexpr bool HOISTM = !(CB == 4 && EPIC >= 0 && (EPIC & LF_EPI_MASKBN) != 0 && (EPIC & LF_EPI_STATS_XHAT) != 0);
            f32x4 bs[NOBIAS ? 1 : NT], hsc[HOISTM ? NT : 1], hsh[HOISTM ? NT : 1];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const unsigned co = (unsigned)(cobw + n * 16 + kq * 4);
                if constexpr (!NOBIAS) bs[n] = a.bias ? ldb4(r_bias, co * 4u, 0u) : zero4();
                if constexpr (HOISTM) { if (epi & LF_EPI_MASKBN) { hsc[n] = ldb4(r_msc, co * 4u, 0u); hsh[n] = ldb4(r_msh, co * 4u, 0u); } }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned dbase = (unsigned)(((G.n[m] * g.Hd + G.i[m]) * g.Wd + G.j[m] + pl) * g.d_pix + g.d_choff + cobw + kq * 4);
                f32x4 la[NT], lm[NT], lx[NT], ld[NT];
                auto tile_ld = [&](const unsigned char* tile, int n) __attribute__((always_inline)) {      // this lane's 4 channels of (m, n)
                    const int p = m * 16 + pl, chunk = (cobw + n * 16 + kq * 4) >> 3;
                    const uint2 q = *reinterpret_cast<const uint2*>(tile + p * PIXB + ((chunk ^ p) & (NCH - 1)) * 16 + (kq & 1) * 8);
                    f32x4 v;
                    v.x = __uint_as_float(q.x << 16); v.y = __uint_as_float(q.x & 0xffff0000u);
                    v.z = __uint_as_float(q.y << 16); v.w = __uint_as_float(q.y & 0xffff0000u);
                    return v;
                };
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (epi & LF_EPI_ADD) la[n] = C::ST_ADD ? tile_ld(otile, n) : epi_ld<S16>(r_add, dbase + n * 16);
                    if (epi & LF_EPI_MASK) lm[n] = C::ST_MSK ? tile_ld(mtile, n) : epi_ld<S16>(r_msk, dbase + n * 16);
                    if (epi & (LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) lx[n] = C::ST_AUX ? tile_ld(xtile, n) : epi_ld<S16>(r_aux, dbase + n * 16);
                    if ((epi & LF_EPI_STATS_XHAT) && a.dm) ld[n] = ldb4(r_dm, (unsigned)(G.n[m] * g.Cd + cobw + n * 16 + kq * 4) * 4u, 0u);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 v = acc[n][m];
                    if constexpr (!NOBIAS) v += bs[n];
                    if (epi & LF_EPI_ADD) v += la[n];
                    if (epi & LF_EPI_MASK) v = keep_pos(v, lm[n]);
                    if (epi & LF_EPI_MASKBN) {
                        const unsigned co = (unsigned)(cobw + n * 16 + kq * 4);
                        v = keep_pos(v, lx[n] * (HOISTM ? hsc[HOISTM ? n : 0] : ldb4(r_msc, co * 4u, 0u)) + (HOISTM ? hsh[HOISTM ? n : 0] : ldb4(r_msh, co * 4u, 0u)));
                    }
                    if (epi & LF_EPI_RELU) v = max0(v);
                    lf_bf16x4 b;
                    b[0] = (lf_bf16)v.x; b[1] = (lf_bf16)v.y; b[2] = (lf_bf16)v.z; b[3] = (lf_bf16)v.w;
                    // output tile: pixel p = m*16 + pl, 16-byte chunk (channel / 8) XOR p, half kq & 1
                    const int p = m * 16 + pl, chunk = (cobw + n * 16 + kq * 4) >> 3;
                    *reinterpret_cast<lf_bf16x4*>(otile + p * PIXB + ((chunk ^ p) & (NCH - 1)) * 16 + (kq & 1) * 8) = b;
                    if (stats) {
                        v.x = (float)b[0]; v.y = (float)b[1]; v.z = (float)b[2]; v.w = (float)b[3];     // statistics of the values as stored
                        if (!G.ok[m]) v = zero4();
                        if (epi & LF_EPI_STATS_SQ) {
                            if (PIVOT && sub == 0 && m == 0) piv[PIVOT ? n : 0] = bcast16(v);      // the item's first pixel (this wave owns all 256 of its channels)
                            f32x4 dv = PIVOT ? v - piv[PIVOT ? n : 0] : v;
                            if (!G.ok[m]) dv = zero4();
                            s1[n] += dv; s2[n] += dv * dv;
                        }
                        if (epi & LF_EPI_STATS_XHAT) {
                            const f32x4 gm = a.dm ? v * ld[n] : v;
                            s1[n] += gm; s2[n] += gm * lx[n];
                        }
                    }
                }
                if constexpr (!HOISTM) __builtin_amdgcn_sched_barrier(0);      // one pixel tile's operands at a time
            }
            __syncthreads();                          // the output tile is complete
            // whole-line stores: instruction ii of the tile = bytes ii*1024 .. +1023 of the (pixel-major) tile
            constexpr int NI = C::OUT_BYTES / 1024 / WG_WAVES;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int ii = wave * NI + k;
                const int p = (ii * 1024) / PIXB + (lane * 16) / PIXB;            // pixel of the sub-tile
                const int slot = ((lane * 16) % PIXB) / 16;
                const int grp = (ii * 1024 / PIXB) >> 4;                          // uniform: an instruction's pixels lie in one group
                const int gn = grp == 0 ? G.n[0] : grp == 1 ? G.n[1] : grp == 2 ? G.n[2] : G.n[3];
                const int gi = grp == 0 ? G.i[0] : grp == 1 ? G.i[1] : grp == 2 ? G.i[2] : G.i[3];
                const int gj = grp == 0 ? G.j[0] : grp == 1 ? G.j[1] : grp == 2 ? G.j[2] : G.j[3];
                const bool gok = grp == 0 ? G.ok[0] : grp == 1 ? G.ok[1] : grp == 2 ? G.ok[2] : G.ok[3];
                if (gok) {
                    const u32x4v v = *reinterpret_cast<const u32x4v*>(otile + ii * 1024 + lane * 16);
                    const unsigned off = (unsigned)((((gn * g.Hd + gi) * g.Wd + gj + (p & 15)) * g.d_pix + g.d_choff) * 2 + ((slot ^ p) & (NCH - 1)) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v, r_dst, (int)off, 0, 0);
                }
            }
            // (the next sub-tile's first barrier orders these reads of the output tile before its next writes)
        }
        if (stats) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 r1, r2;
                r1.x = sum16(s1[n].x); r1.y = sum16(s1[n].y); r1.z = sum16(s1[n].z); r1.w = sum16(s1[n].w);
                r2.x = sum16(s2[n].x); r2.y = sum16(s2[n].y); r2.z = sum16(s2[n].z); r2.w = sum16(s2[n].w);
                if (epi & LF_EPI_STATS_SQ) {      // pivot sums -> (sum v, M2 about the item's own mean) of the item's valid pixels
                    const unsigned t0 = bx * (unsigned)PIX_PER_WG;
                    const float ni = t0 < npix ? (float)min(npix - t0, (unsigned)PIX_PER_WG) : 0.f;
                    const float rn = ni > 0.f ? 1.f / ni : 0.f;
                    r2 = r2 - r1 * r1 * rn;
                    r2.x = fmaxf(r2.x, 0.f); r2.y = fmaxf(r2.y, 0.f); r2.z = fmaxf(r2.z, 0.f); r2.w = fmaxf(r2.w, 0.f);
                    if (PIVOT) r1 = r1 + piv[PIVOT ? n : 0] * ni;
                }
                if (pl == 0) {            // channel-major rows: [2][Cd][stats_ld] (32-bit indices: 2 * Cd * stats_ld floats is far below 2^31)
                    float* d = a.stats + (unsigned)((cobw + n * 16 + kq * 4) * a.stats_ld) + bx;
                    const unsigned half = (unsigned)(g.Cd * a.stats_ld);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d[(unsigned)(e * a.stats_ld)] = r1[e];
                        d[half + (unsigned)(e * a.stats_ld)] = r2[e];
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing (dead) steps: nothing may land in LDS after this
    __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------
// tapgemm_bf16_wv_kernel ("wave-private", round 6): the 3-tap bf16 convolutions of non_bottleneck_1d at 64 channels.
//
// At 64 channels the whole-line kernel above shares a 64-pixel sub-tile between its four waves (each owns 16 output channels):
// 24 MFMAs per wave between five workgroup barriers, and the launch sits at 4.0 TB/s of tensor bytes where the same kernel at 128
// channels has twice the work per barrier.  But at 64 channels ALL the weights of the convolution are 3 taps x 64 x 64 x 2 B =
// 24 KB = 96 registers per lane -- what a wave of the 128-channel form holds for its quarter of the output channels.  So here a
// wave owns its pixels AND every output channel: wave tile = 32 pixels x 64 channels (8 accumulator tiles), the weights live in
// its registers for the workgroup's life, its operands arrive by LDS-DMA in a PRIVATE ring (stage = one tap of its 32 pixels = 4 KB,
// 4 instructions of 8 pixels x 128 B: whole lines), its output leaves through a private 4 KB tile as whole-line stores -- and no
// instruction of the K loop or the epilogue waits for another wave: the only barriers are the two that merge the four waves'
// BatchNorm partial sums into the 256-pixel statistics row, in the launches that take statistics.
//   ring order (per wave):   s_waitcnt vmcnt(N) lgkmcnt(0)      my DMA of this step has landed; my fragment reads of the step before
//                                                               have retired (their stage is restaged next)
//                            issue the step R - 1 ahead; read fragments; 16 MFMAs
// N counts the vector-memory instructions YOUNGER than the awaited DMA: 4 (R - 2) ring instructions, the item's staging
// instructions behind its first step, and -- vmcnt retires loads AND stores in issue order on gfx9-family hardware -- the four
// whole-line stores of the previous item's epilogue in front of an item's first step (all four are always issued: a group beyond
// the tensor stores to the out-of-range offset, which the buffer unit drops).
// Work: a workgroup walks 256-pixel tiles (the statistics rows of the other kernels) inside its XCD's range; wave w takes pixels
// w*64 .. w*64+63 of the tile as two items of 32.  K order = tapgemm_bf16_kernel's (tap, then channel): results bit-identical.
// Measured on one box (tools/bf16_ab.py, config 3's 64 x 80 x 160 x 64 launches, whole-line kernel -> this one): forward 52.1 -> 41.6 us,
// data gradient + mask 65.0 -> 60.3 / 66.7 -> 58.1.  The BN+ReLU operand prologue (the block's third convolution) was built here
// too -- the wave transforming the stage it brought in, in place, behind its own vmcnt wait -- and is SLOWER than the streaming
// kernel's register form (102.9 vs 88.3 us: the transform sits between the wave's DMA wait and its fragment reads, three times per
// pixel, with one partner wave to hide it): removed, those launches stay on tapgemm_bf16_kernel.
// ---------------------------------------------------------------------------------------
constexpr int WV_PX = 32, WV_STAGE = WV_PX * 128;
template <int EPIC> struct WvCfg {
    static constexpr bool ST_ADD = EPIC >= 0 && (EPIC & LF_EPI_ADD) != 0, ST_MSK = EPIC >= 0 && (EPIC & LF_EPI_MASK) != 0,
                          ST_AUX = EPIC >= 0 && (EPIC & (LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) != 0;
    static constexpr int NBUF = (ST_MSK ? 1 : 0) + (ST_AUX ? 1 : 0);                  // tiles beside the output tile
    static constexpr int R = NBUF == 1 ? 3 : 4;                                       // ring stages (one extra tile: 3 keeps two workgroups per CU)
    static constexpr int NAUXI = ((ST_ADD ? 1 : 0) + NBUF) * 4;                       // staging instructions per item
    static constexpr int WAVE_LDS = (R + 1 + NBUF) * WV_STAGE;
    static constexpr size_t LDS = (size_t)WG_WAVES * WAVE_LDS;
};

// (two workgroups per CU by LDS except with two staged operand tiles or run-time flags: those get the whole register file)
template <int EPIC>
__global__ __launch_bounds__(256, (EPIC >= 0 && WvCfg<EPIC>::NBUF < 2) ? 2 : 1) void tapgemm_bf16_wv_kernel(const LfTapGeom g, const LfTapArgs a, const int pro, const int epi_rt) {
    typedef WvCfg<EPIC> C;
    constexpr int NT = 4, MW = 2, R = C::R, CD = 64;
    constexpr bool S16 = true;
    const int epi = EPIC >= 0 ? EPIC : epi_rt;
    constexpr bool NOBIAS = EPIC >= 0 && (EPIC & (LF_EPI_MASK | LF_EPI_ADD | LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) != 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pl = lane & 15, kq = lane >> 4;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl), ngroups = npix >> 4, GR = (unsigned)g.Wl >> 4;
    const unsigned ntiles = (npix + PIX_PER_WG - 1) / PIX_PER_WG;
    unsigned it_first = blockIdx.x, it_stride = gridDim.x, it_end = ntiles;
    if ((gridDim.x & 7u) == 0 && (ntiles & 7u) == 0) {
        const unsigned per = ntiles >> 3;
        it_first = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); it_stride = gridDim.x >> 3; it_end = (blockIdx.x & 7u) * per + per;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lf_tap_lds + (unsigned)(wave * C::WAVE_LDS);
    unsigned char* const ring = lf_tap_lds + wave * C::WAVE_LDS;
    unsigned char* const otile = ring + R * WV_STAGE;
    unsigned char* const mtile = otile + WV_STAGE;                                     // mask source (when staged)
    unsigned char* const xtile = otile + WV_STAGE * (C::ST_MSK ? 2 : 1);              // BN-backward operand (when staged)

    // ---- all the weights -> registers: [tap][32-channel k-block][16-channel output tile], the MFMA A operand
    bf16x8 wr[3][2][NT];
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wp16, 0xffffffffu);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    wr[t][kb][n] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rw, (int)((((t * 2 + kb) * 4 + kq) * CD + n * 16 + pl) * 16), 0, 0));
    }
    int tapv = 0;            // tap t's offsets in lane t (v_readlane: see tapgemm_bf16_wl_kernel)
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (lane == t) tapv = (g.tdh[t] & 0xffff) | (g.tdw[t] << 16);
    const i32x4s rx = make_rsrc_words(a.src, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 2, (long)LF_OOB));
    const int spix2 = g.s_pix * 2;
    // DMA mapping: instruction q of a step carries pixels q*8 .. q*8+7 of the item (group q >> 1): lane -> pixel q*8 + (lane >> 3),
    // LDS slot lane & 7 = chunk XOR ((pixel >> 1) & 7) -- which depends on q & 1 only
    const int dpx = lane >> 3;
    unsigned lane_x[2];
    int lane_dx[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int chunk = (lane & 7) ^ ((jj * 4 + (lane >> 4)) & 7);
        lane_x[jj] = (unsigned)(((jj * 8 + dpx) * g.s_pix + g.s_choff) * 2 + chunk * 16);
        lane_dx[jj] = jj * 8 + dpx;
    }
    // the two 16-pixel groups of item (tile, h) of this wave: (image, row, first column), wave-uniform
    struct Item { int n[2], i[2], j[2]; bool ok[2]; };
    auto item_at = [&](unsigned tile, int h, Item& o) __attribute__((always_inline)) {
        const unsigned G0 = tile * 16u + (unsigned)(wave * 4 + h * 2);
        const unsigned Gc = G0 < ngroups ? G0 : 0u;
        const unsigned Rw = Gc / GR;
        int jg = __builtin_amdgcn_readfirstlane((int)(Gc - Rw * GR));
        int n = __builtin_amdgcn_readfirstlane((int)(Rw / (unsigned)g.Hl));
        int i = __builtin_amdgcn_readfirstlane((int)Rw) - n * g.Hl;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            o.ok[q] = G0 + q < ngroups;
            o.n[q] = n; o.i[q] = i; o.j[q] = jg * 16;
            if (++jg == (int)GR) { jg = 0; if (++i == g.Hl) { i = 0; ++n; } }
        }
    };
    // ---- load cursor (wave-uniform): tile, half, tap, ring stage; its item's groups as (pixel index without the tap, row, column)
    unsigned tile_ld = it_first;
    int h_ld = 0, t_ld = 0, st_ld = 0;
    int lrow[2] = {0, 0}, liy[2] = {0, 0}, lsx[2] = {0, 0};
    bool lok[2] = {false, false};
    auto cursor_item = [&]() __attribute__((always_inline)) {
        lok[0] = false; lok[1] = false;
        if (tile_ld < it_end) {
            Item I;
            item_at(tile_ld, h_ld, I);
#pragma unroll
            for (int q = 0; q < 2; ++q) { lok[q] = I.ok[q]; liy[q] = I.i[q]; lsx[q] = I.j[q]; lrow[q] = (I.n[q] * g.Hs + I.i[q]) * g.Ws + I.j[q]; }
        }
    };
    cursor_item();
    auto issue = [&]() __attribute__((always_inline)) {
        const int tv = __builtin_amdgcn_readlane(tapv, t_ld);
        const int dh = (int)(short)(tv & 0xffff), dw = tv >> 16;
        const unsigned dst = lds0 + (unsigned)(st_ld * WV_STAGE);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gq = q >> 1, jj = q & 1;
            const int sy = liy[gq] + dh;
            const bool yok = lok[gq] && sy >= 0 && sy < g.Hs;
            const unsigned base = (unsigned)((lrow[gq] + dh * g.Ws + dw) * spix2);
            const bool in = yok && (unsigned)(lsx[gq] + dw + lane_dx[jj]) < (unsigned)g.Ws;
            lds_dma16(rx, dst + (unsigned)q * 1024u, in ? base + lane_x[jj] : LF_OOB, 0u);
        }
        st_ld = st_ld == R - 1 ? 0 : st_ld + 1;
        if (++t_ld == 3) {
            t_ld = 0;
            if (++h_ld == 2) { h_ld = 0; tile_ld += it_stride; }
            cursor_item();
        }
    };
#pragma unroll
    for (int i = 0; i < R - 1; ++i) issue();

    // ---- staging of the epilogue's operand tensors (whole lines, the output tile's layout): instruction ii = pixels ii*8 .. +7 of
    // the item: lane -> pixel p = ii*8 + (lane >> 3), 16-byte slot lane & 7 = chunk XOR (p & 7)
    const unsigned dbytes = (unsigned)min((long)g.N * g.Hd * g.Wd * g.d_pix * 2, (long)LF_OOB);
    const i32x4s ra_add = make_rsrc_words(a.add_src, dbytes), ra_msk = make_rsrc_words(a.mask_src, dbytes), ra_aux = make_rsrc_words(a.aux, dbytes);
    auto tile_lane_off = [&](int ii) __attribute__((always_inline)) {          // byte offset of this lane's chunk inside its group's 16 pixels
        const int p = ii * 8 + (lane >> 3);
        return (unsigned)((((p & 15) * g.d_pix + g.d_choff) * 2) + (((lane & 7) ^ p) & 7) * 16);
    };
    auto stage_tensors = [&](const Item& I) __attribute__((always_inline)) {
        if constexpr (C::NAUXI > 0) {
            const unsigned dst0 = lds0 + (unsigned)(R * WV_STAGE);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int gq = ii >> 1;
                const unsigned base = (unsigned)(((I.n[gq] * g.Hd + I.i[gq]) * g.Wd + I.j[gq]) * g.d_pix * 2);
                const unsigned vo = I.ok[gq] ? base + tile_lane_off(ii) : LF_OOB;      // (a group beyond the tensor: zeros)
                if constexpr (C::ST_ADD) lds_dma16(ra_add, dst0 + (unsigned)(ii * 1024), vo, 0u);
                if constexpr (C::ST_MSK) lds_dma16(ra_msk, dst0 + (unsigned)(WV_STAGE + ii * 1024), vo, 0u);
                if constexpr (C::ST_AUX) lds_dma16(ra_aux, dst0 + (unsigned)(WV_STAGE * (C::ST_MSK ? 2 : 1) + ii * 1024), vo, 0u);
            }
        }
    };

    // epilogue constants (destination bounded by the tensor: the store of a group beyond it goes to the out-of-range offset and is dropped)
    const __amdgpu_buffer_rsrc_t r_dst = make_rsrc(a.dst, dbytes), r_add = make_rsrc(a.add_src, 0xffffffffu),
                                 r_msk = make_rsrc(a.mask_src, 0xffffffffu), r_aux = make_rsrc(a.aux, 0xffffffffu),
                                 r_bias = make_rsrc(a.bias, 0xffffffffu), r_msc = make_rsrc(a.msc, 0xffffffffu),
                                 r_msh = make_rsrc(a.msh, 0xffffffffu), r_dm = make_rsrc(a.dm, 0xffffffffu);
    // fragment read addresses inside a stage: pixel m*16 + pl (128 B rows), chunk kb*4 + kq XOR (pl >> 1) & 7
    const unsigned xf0 = (unsigned)(pl * 128 + (((0 + kq) ^ (pl >> 1)) & 7) * 16), xf1 = (unsigned)(pl * 128 + (((4 + kq) ^ (pl >> 1)) & 7) * 16);
    const bool stats = (epi & (LF_EPI_STATS_SQ | LF_EPI_STATS_XHAT)) != 0;
    int st_c = 0;
    bool first_item = true;          // no stores in flight in front of the very first step
    for (unsigned bx = it_first; bx < it_end; bx += it_stride) {
        constexpr bool PIVOT = EPIC >= 0;    // the pivot of the BN forward sums (see LF_TAPGEMM_EPILOGUE); the run-time-flag form sums about 0
        f32x4 s1[NT], s2[NT], piv[PIVOT ? NT : 1];
#pragma unroll
        for (int n = 0; n < NT; ++n) { s1[n] = zero4(); s2[n] = zero4(); piv[PIVOT ? n : 0] = zero4(); }
        for (int h = 0; h < 2; ++h) {
            f32x4 acc[NT][MW];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MW; ++m) acc[n][m] = zero4();
            Item I;
            item_at(bx, h, I);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                // my DMA of this step has landed; my fragment reads of the step before have retired (header: what N counts)
                if (t == 0) {
                    if (first_item) wait_vm_lgkm0<4 * (R - 2)>();
                    else wait_vm_lgkm0<4 * (R - 2) + 4>();
                } else wait_vm_lgkm0<4 * (R - 2) + C::NAUXI>();
                issue();                             // R - 1 steps ahead -> the stage the previous step occupied
                if (t == 0) stage_tensors(I);
                const unsigned char* st = ring + st_c * WV_STAGE;
                st_c = st_c == R - 1 ? 0 : st_c + 1;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8 xb[MW];
#pragma unroll
                    for (int m = 0; m < MW; ++m) xb[m] = *reinterpret_cast<const bf16x8*>(st + (kb ? xf1 : xf0) + m * 2048);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[t][kb][n], xb[m], acc[n][m], 0, 0, 0);
                }
            }
            first_item = false;
            // ---- epilogue of the item: accumulator tile (n, m) = channels n*16 + kq*4 .. +3 of pixel m*16 + pl
            if constexpr (C::NAUXI > 0) wait_vm_lgkm0<8>();      // the staged tensors have landed (the two ring steps issued since are younger)
            constexpr bool HOISTM = !(EPIC >= 0 && (EPIC & LF_EPI_MASKBN) != 0 && (EPIC & LF_EPI_STATS_XHAT) != 0);
            f32x4 bs[NOBIAS ? 1 : NT], hsc[HOISTM ? NT : 1], hsh[HOISTM ? NT : 1];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const unsigned co = (unsigned)(n * 16 + kq * 4);
                if constexpr (!NOBIAS) bs[n] = a.bias ? ldb4(r_bias, co * 4u, 0u) : zero4();
                if constexpr (HOISTM) { if (epi & LF_EPI_MASKBN) { hsc[n] = ldb4(r_msc, co * 4u, 0u); hsh[n] = ldb4(r_msh, co * 4u, 0u); } }
            }
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const unsigned dbase = (unsigned)(((I.n[m] * g.Hd + I.i[m]) * g.Wd + I.j[m] + pl) * g.d_pix + g.d_choff + kq * 4);
                auto tile_ld = [&](const unsigned char* tile, int n) __attribute__((always_inline)) {      // this lane's 4 channels of (m, n)
                    const int p = m * 16 + pl, chunk = (n * 16 + kq * 4) >> 3;
                    const uint2 q = *reinterpret_cast<const uint2*>(tile + p * 128 + ((chunk ^ p) & 7) * 16 + (kq & 1) * 8);
                    f32x4 v;
                    v.x = __uint_as_float(q.x << 16); v.y = __uint_as_float(q.x & 0xffff0000u);
                    v.z = __uint_as_float(q.y << 16); v.w = __uint_as_float(q.y & 0xffff0000u);
                    return v;
                };
                // (two channel tiles at a time: with all four tiles' operands live beside the 96 weight registers the three-operand
                // epilogues spilled 10-31 registers)
#pragma unroll
                for (int n0 = 0; n0 < NT; n0 += 2) {
                f32x4 la[NT], lm[NT], lx[NT];
#pragma unroll
                for (int n = n0; n < n0 + 2; ++n) {
                    if (epi & LF_EPI_ADD) la[n] = C::ST_ADD ? tile_ld(otile, n) : epi_ld<S16>(r_add, dbase + n * 16);
                    if (epi & LF_EPI_MASK) lm[n] = C::ST_MSK ? tile_ld(mtile, n) : epi_ld<S16>(r_msk, dbase + n * 16);
                    if (epi & (LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) lx[n] = C::ST_AUX ? tile_ld(xtile, n) : epi_ld<S16>(r_aux, dbase + n * 16);
                }
#pragma unroll
                for (int n = n0; n < n0 + 2; ++n) {
                    f32x4 v = acc[n][m];
                    if constexpr (!NOBIAS) v += bs[n];
                    if (epi & LF_EPI_ADD) v += la[n];
                    if (epi & LF_EPI_MASK) v = keep_pos(v, lm[n]);
                    if (epi & LF_EPI_MASKBN) {
                        const unsigned co = (unsigned)(n * 16 + kq * 4);
                        v = keep_pos(v, lx[n] * (HOISTM ? hsc[HOISTM ? n : 0] : ldb4(r_msc, co * 4u, 0u)) + (HOISTM ? hsh[HOISTM ? n : 0] : ldb4(r_msh, co * 4u, 0u)));
                    }
                    if (epi & LF_EPI_RELU) v = max0(v);
                    lf_bf16x4 b;
                    b[0] = (lf_bf16)v.x; b[1] = (lf_bf16)v.y; b[2] = (lf_bf16)v.z; b[3] = (lf_bf16)v.w;
                    // output tile: pixel p = m*16 + pl, 16-byte chunk (channel / 8) XOR p, half kq & 1
                    const int p = m * 16 + pl, chunk = (n * 16 + kq * 4) >> 3;
                    *reinterpret_cast<lf_bf16x4*>(otile + p * 128 + ((chunk ^ p) & 7) * 16 + (kq & 1) * 8) = b;
                    if (stats) {
                        v.x = (float)b[0]; v.y = (float)b[1]; v.z = (float)b[2]; v.w = (float)b[3];     // statistics of the values as stored
                        if (!I.ok[m]) v = zero4();
                        if (epi & LF_EPI_STATS_SQ) {
                            if (PIVOT && h == 0 && m == 0) piv[PIVOT ? n : 0] = bcast16(v);      // this wave's first pixel of the tile
                            f32x4 dv = PIVOT ? v - piv[PIVOT ? n : 0] : v;
                            if (!I.ok[m]) dv = zero4();
                            s1[n] += dv; s2[n] += dv * dv;
                        }
                        if (epi & LF_EPI_STATS_XHAT) {
                            // (the Dropout2d factor of (image, channel): an L1-resident vector, read where it is used instead of held)
                            const f32x4 gm = a.dm ? v * ldb4(r_dm, (unsigned)(I.n[m] * g.Cd + n * 16 + kq * 4) * 4u, 0u) : v;
                            s1[n] += gm; s2[n] += gm * lx[n];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // one chunk's operands at a time
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my quads are in the output tile
            // whole-line stores: instruction ii = pixels ii*8 .. +7 of the item (all four always issued: header)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int gq = ii >> 1;
                const u32x4v v = *reinterpret_cast<const u32x4v*>(otile + ii * 1024 + lane * 16);
                const unsigned base = (unsigned)(((I.n[gq] * g.Hd + I.i[gq]) * g.Wd + I.j[gq]) * g.d_pix * 2);
                __builtin_amdgcn_raw_buffer_store_b128(v, r_dst, (int)(I.ok[gq] ? base + tile_lane_off(ii) : LF_OOB), 0, 0);
            }
            // (the next item's first wait carries lgkmcnt(0): these reads of the output tile retire before its next writes)
        }
        if (stats) {
            // the four waves' sums of this 256-pixel tile -> one partial row (the merge of LF_TAPGEMM_EPILOGUE); staging area = the
            // first 512 bytes of every wave's output tile (free: its stores read it above, the next item writes it 3 steps from now)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float (*sred)[4][8] = reinterpret_cast<float (*)[4][8]>(otile);           // [n][kq][8]
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 r1, r2;
                r1.x = sum16(s1[n].x); r1.y = sum16(s1[n].y); r1.z = sum16(s1[n].z); r1.w = sum16(s1[n].w);
                r2.x = sum16(s2[n].x); r2.y = sum16(s2[n].y); r2.z = sum16(s2[n].z); r2.w = sum16(s2[n].w);
                if (epi & LF_EPI_STATS_SQ) {      // pivot sums -> (sum v, M2 about the wave's own mean) of the wave's nw valid pixels
                    const unsigned t0 = (bx * (unsigned)WG_WAVES + (unsigned)wave) * 64u;
                    const float nw = t0 < npix ? (float)min(npix - t0, 64u) : 0.f;
                    const float rn = nw > 0.f ? 1.f / nw : 0.f;
                    r2 = r2 - r1 * r1 * rn;
                    r2.x = fmaxf(r2.x, 0.f); r2.y = fmaxf(r2.y, 0.f); r2.z = fmaxf(r2.z, 0.f); r2.w = fmaxf(r2.w, 0.f);
                    if (PIVOT) r1 = r1 + piv[PIVOT ? n : 0] * nw;
                }
                if (pl == 0) {
                    float* d = sred[n][kq];
                    d[0] = r1.x; d[1] = r1.y; d[2] = r1.z; d[3] = r1.w; d[4] = r2.x; d[5] = r2.y; d[6] = r2.z; d[7] = r2.w;
                }
            }
            __syncthreads();
            if (threadIdx.x < NT * 4 * 8) {
                const int tt = threadIdx.x;
                const int j = tt & 7, q = (tt >> 3) & 3, n = tt >> 5;
                auto at = [&](int w, int jj) { return reinterpret_cast<const float (*)[4][8]>(lf_tap_lds + w * C::WAVE_LDS + R * WV_STAGE)[n][q][jj]; };
                float v = at(0, j) + at(1, j) + at(2, j) + at(3, j);
                if ((epi & LF_EPI_STATS_SQ) && j >= 4) {      // Chan: M2 = sum_w M2_w + sum_w n_w (mean_w - mean)^2
                    float nwv[4], sw[4], nt = 0.f, stt = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned t0w = (bx * (unsigned)WG_WAVES + (unsigned)w) * 64u;
                        nwv[w] = t0w < npix ? (float)min(npix - t0w, 64u) : 0.f;
                        sw[w] = at(w, j - 4);
                        nt += nwv[w]; stt += sw[w];
                    }
                    const float mg = nt > 0.f ? stt / nt : 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (nwv[w] > 0.f) { const float dmw = sw[w] / nwv[w] - mg; v += nwv[w] * dmw * dmw; }
                }
                a.stats[((long)(j >> 2) * g.Cd + n * 16 + q * 4 + (j & 3)) * a.stats_ld + bx] = v;      // channel-major rows
            }
            __syncthreads();      // the staging areas are output tiles again
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing (dead) steps: nothing may land in LDS after this
    __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------
// tapgemm_bf16_lean_kernel: the 16 -> 16 channel 3-tap convolutions on bf16 tensors (decoder, full-resolution stage: HBM-bound,
// a pixel is 32 bytes).  The general bf16 kernel ran them in its run-time-flag form, one 32-channel K-step per tap with half
// of every operand register zero padding and one step in flight (101 us per launch at 160 x 320 x 64 images, 2.1 TB/s).  Here
//  * one MFMA contracts TWO taps: k-blocks 0-1 of the K = 32 step are tap 2i's 16 channels, k-blocks 2-3 tap 2i+1's -- a lane's
//    16-byte load is half a pixel at the tap its k-block belongs to, an instruction covers 2 x 16 consecutive pixels x 32 B
//    (whole lines); the three taps are 2 MFMAs per 16 x 16 tile, all 8 loads of a wave requested up front;
//  * prologue and epilogue flags compiled in, (image, row) of the 16-pixel groups in scalar registers (Wl % 16 == 0).
// The chain order (tap 0 + tap 1 in one K = 32 dot product, then tap 2) differs from tapgemm_bf16_kernel's three K-steps:
// results agree to fp32 rounding, not bit for bit.
// ---------------------------------------------------------------------------------------
template <int PROC, int EPIC>
__global__ __launch_bounds__(256, 4) void tapgemm_bf16_lean_kernel(const LfTapGeom g, const LfTapArgs a, const int pro_rt, const int epi_rt) {
    constexpr int NT = 1;
    constexpr bool S16 = true, HOISTV = EPIC >= 0;
    const int epi = EPIC >= 0 ? EPIC : epi_rt;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pl = lane & 15, kq = lane >> 4;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl), ngroups = npix >> 4, GR = (unsigned)g.Wl >> 4;
    const int cob = 0;
    unsigned bx = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    RingGroups G;
    ring_groups(bx, wave, g.Hl, GR, ngroups, G);
    const int hi = kq >> 1, c8 = (kq & 1) * 8;              // which tap of the pair, which half of the pixel
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wp16, 0xffffffffu),
                                 rx = make_rsrc(a.src, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 2, (long)LF_OOB));
    // weights: packed [tap][4 k-blocks of 8 channels (2 real, 2 zero)][Cd][8]; MFMA i, lane (pl, kq): tap 2i + hi, k-block kq & 1
    bf16x8 wb[2];
    wb[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)((((hi ? 1 : 0) * 4 + (kq & 1)) * g.Cd + pl) * 16), 0, 0));
    {
        u32x4v w2 = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(((2 * 4 + (kq & 1)) * g.Cd + pl) * 16), 0, 0);
        if (hi) { w2[0] = 0u; w2[1] = 0u; w2[2] = 0u; w2[3] = 0u; }       // there is no fourth tap
        wb[1] = __builtin_bit_cast(bf16x8, w2);
    }
    (void)pro_rt;
    u32x4v xr[2][MT];
    bool in[2][MT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int dh = i == 0 ? (hi ? g.tdh[1] : g.tdh[0]) : g.tdh[2], dw = i == 0 ? (hi ? g.tdw[1] : g.tdw[0]) : g.tdw[2];
        const bool tapok = i == 0 || !hi;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int sy = G.i[m] * g.ssh + dh, sx = (G.j[m] + pl) * g.ssw + dw;
            in[i][m] = tapok && G.ok[m] && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
            const unsigned off = (unsigned)((((G.n[m] * g.Hs + sy) * g.Ws + sx) * g.s_pix + g.s_choff + c8) * 2);
            xr[i][m] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(in[i][m] ? off : LF_OOB), 0, 0);
        }
    }
    f32x4 sc0, sc1, sh0, sh1;
    if constexpr (PROC == LF_PRO_BNRELU) {
        sc0 = ldg4(a.pro_sc + c8); sc1 = ldg4(a.pro_sc + c8 + 4);
        sh0 = ldg4(a.pro_sh + c8); sh1 = ldg4(a.pro_sh + c8 + 4);
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bf16x8 xb;
            if constexpr (PROC == LF_PRO_BNRELU) {          // relu(bn(x)) in fp32 on the widened values; padding is zero AFTER the transform
                const u32x4v r = xr[i][m];
                f32x4 lo, hv;
                lo.x = __uint_as_float(r[0] << 16); lo.y = __uint_as_float(r[0] & 0xffff0000u);
                lo.z = __uint_as_float(r[1] << 16); lo.w = __uint_as_float(r[1] & 0xffff0000u);
                hv.x = __uint_as_float(r[2] << 16); hv.y = __uint_as_float(r[2] & 0xffff0000u);
                hv.z = __uint_as_float(r[3] << 16); hv.w = __uint_as_float(r[3] & 0xffff0000u);
                lo = max0(lo * sc0 + sh0); hv = max0(hv * sc1 + sh1);
                const bool ok = in[i][m];
                lo.x = ok ? lo.x : 0.f; lo.y = ok ? lo.y : 0.f; lo.z = ok ? lo.z : 0.f; lo.w = ok ? lo.w : 0.f;
                hv.x = ok ? hv.x : 0.f; hv.y = ok ? hv.y : 0.f; hv.z = ok ? hv.z : 0.f; hv.w = ok ? hv.w : 0.f;
                xb = cvt_bf16x8(lo, hv);
            } else xb = __builtin_bit_cast(bf16x8, xr[i][m]);
            acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i], xb, i ? acc[0][m] : zero4(), 0, 0, 0);
        }
    int pn[MT], pi[MT], pj[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { pn[m] = G.n[m]; pi[m] = G.i[m]; pj[m] = G.j[m] + pl; pv[m] = G.ok[m]; }
    LF_TAPGEMM_EPILOGUE
}

// ---------------------------------------------------------------------------------------
// fp32 ON THE bf16 MATRIX CORES ("split" mode, LfTapArgs::split = 9).  gfx950 multiplies bf16 16x faster
// than fp32 (v_mfma_f32_16x16x32_bf16: 16 cycles for K = 32; v_mfma_f32_16x16x4_f32: 8 x 32 cycles for the same K),
// so an fp32 product is formed from bf16 pieces instead: every fp32 operand is split EXACTLY into three bf16 values
//     a = a_h + a_m + a_l,   a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m)      (8 + 8 + 8 mantissa bits)
// (both subtractions are exact in fp32, a_l is exact because at most 8 significant bits are left), and
//     a * b = sum_{i,j} a_i * b_j
// where each of the nine bf16 x bf16 products is exact in the fp32 accumulator; all nine are kept: the result is an fp32
// accumulation of exact products, the same contract as the fp32 FMA chain of tapgemm_kernel.  (A 6-term form that dropped
// a_m*b_l, a_l*b_m, a_l*b_l -- < 2^-24 |a b| together -- existed through round 5; no BASELINE configuration used it.)
// Tensors, accumulators, epilogue and statistics are fp32 as in mode 0.
// Cost per 64 x 64 x 32 wave step: 144 MFMAs x 16 cycles = 2304 cycles against 4096 on the fp32 cores.
// The pixel operand is split in registers (v_cvt_pk_bf16_f32 + exact residuals, ~5 VALU ops per element, issued in the
// shadow of the MFMAs); weights are split once per forward by the pack kernel ([tap][K/8][Cd][3][8] bf16) and, being
// 3x the bytes of the fp32 form, staged ONCE per workgroup and step through LDS (double-buffered, one barrier per step)
// instead of once per wave from L1, which would saturate the 64 B/clk L1 path at this MFMA rate.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(const f32x4 lo, const f32x4 hi, bf16x8& h, bf16x8& m, bf16x8& l) {
    f32x2 p[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
    u32x4 uh, um, ul;
#pragma unroll
    for (int i = 0; i < 4; ++i) uh[i] = split_pair(p[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) um[i] = split_pair(p[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i)     // what is left has at most 8 significant bits: the third conversion is exact
        ul[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p[i], bf16x2));
    h = __builtin_bit_cast(bf16x8, uh); m = __builtin_bit_cast(bf16x8, um); l = __builtin_bit_cast(bf16x8, ul);
}

// Workgroup = 512 threads = two 4-wave groups A (waves 0-3) and B (waves 4-7); waves w and w+4 share SIMD w.  A wave's
// step has a VALU part (split the 32 pixel values it loaded, issue the next loads) and a matrix part (144 MFMAs).
// Two waves running the same code in lockstep would both want the VALU, then both the matrix pipe; so B runs HALF A STEP
// behind A, phase-locked by the workgroup barrier: while A multiplies, B splits, and vice versa -- the matrix pipe of every
// SIMD always has exactly one wave streaming back-to-back MFMAs.  Group A also stages the weights: W[s+1] is written to
// the idle LDS buffer during A's split phase (B is reading W[s-1]'s successor W[s] from the other one).
#undef LF_EPI_GROUPS
#define LF_EPI_GROUPS 2
template <int NT, int PROC, int EPIC = -1, bool DBG = false>
__global__ __launch_bounds__(512, 2) void tapgemm_split_kernel(const LfTapGeom g, const LfTapArgs a, const int pro, const int epi_rt) {
    const int epi = EPIC >= 0 ? EPIC : epi_rt;      // compiled-in epilogue flags, as in tapgemm_kernel
    constexpr bool S16 = false, HOISTV = EPIC >= 0;
    constexpr int WAVES = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, kq = lane >> 4;
    const int grp = wave >> 2;
    unsigned long long tstamp[4] = {0ull, 0ull, 0ull, 0ull};
    if constexpr (DBG) tstamp[0] = __builtin_amdgcn_s_memrealtime();
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl);
    const int cob = blockIdx.y * NT * 16;
    unsigned bx = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const unsigned tile0 = (bx * WAVES + wave) * (MT * 16);

    int pn[MT], pi[MT], pj[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const unsigned p = tile0 + m * 16 + pl;
        pv[m] = p < npix;
        const unsigned q = pv[m] ? p : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj[m] = (int)(q - r * (unsigned)g.Wl);
        pn[m] = (int)(r / (unsigned)g.Hl);
        pi[m] = (int)(r - (unsigned)pn[m] * (unsigned)g.Hl);
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = zero4();

    struct Raw { f32x4 xl[MT], xh[MT]; f32x4 sc0, sc1, sh0, sh1; unsigned ok; };
    __shared__ uint4 tab_off[WAVES][LF_MAX_TAPS][64];
    __shared__ unsigned tab_ok[WAVES][LF_MAX_TAPS][64];
    __shared__ u32x4 wl[2][3][4][NT * 16];          // [buffer][piece][k-block of 8][output channel]: 16 B rows, conflict-free
    for (int t = 0; t < g.ntaps; ++t) {
        const int dh = g.tdh[t], dw = g.tdw[t];
        unsigned o[MT], okb = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int sy = pi[m] * g.ssh + dh, sx = pj[m] * g.ssw + dw;
            const bool in = pv[m] && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
            const int syc = min(max(sy, 0), g.Hs - 1), sxc = min(max(sx, 0), g.Ws - 1);
            o[m] = (unsigned)(((pn[m] * g.Hs + syc) * g.Ws + sxc) * g.s_pix + g.s_choff + kq * 8);
            okb |= (in ? 1u : 0u) << m;
        }
        tab_off[wave][t][lane] = make_uint4(o[0], o[1], o[2], o[3]);
        tab_ok[wave][t][lane] = okb;
    }
    const int ncb = g.Cs >> 5;                               // 32-channel steps per tap (launcher: Cs % 32 == 0)
    const int nsteps = g.ntaps * ncb;
    const int ntaps = g.ntaps;
    // weight staging by group A: thread -> (k-block, output channel) of the workgroup's NT*16-channel slab
    const int tid = threadIdx.x;
    const bool filler = tid < NT * 64;                       // NT = 4: exactly the 256 threads of group A
    const int f_kb = filler ? tid / (NT * 16) : 0, f_co = filler ? tid % (NT * 16) : 0;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp16) + ((long)f_kb * g.Cd + cob + f_co) * 3;
    const int wstep = g.Cd * 4 * 3;                          // u32x4 per 32-channel step
    int wstepi = 0;
    u32x4 wreg[3];
    auto wfetch = [&]() {
        const u32x4* p = wsrc + (long)min(wstepi, nsteps - 1) * wstep;
        wreg[0] = p[0]; wreg[1] = p[1]; wreg[2] = p[2];
        ++wstepi;
    };
    auto wstore = [&](int buf) {
        if (filler) { wl[buf][0][f_kb][f_co] = wreg[0]; wl[buf][1][f_kb][f_co] = wreg[1]; wl[buf][2][f_kb][f_co] = wreg[2]; }
    };
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.src, 0xffffffffu), rsc = make_rsrc(a.pro_sc, 0xffffffffu),
                                 rsh = make_rsrc(a.pro_sh, 0xffffffffu);
    int t_ld = 0, cb_ld = 0;
    auto issue = [&](Raw& S) {
        const bool live = t_ld < ntaps;
        const int tc = live ? t_ld : ntaps - 1;
        const uint4 o = tab_off[wave][tc][lane];
        const unsigned okb = tab_ok[wave][tc][lane];
        const int c32 = cb_ld * 32;
        const unsigned cs = (unsigned)c32 * 4u;          // scalar byte offset of the channel step (buffer-addressed, see ldb4)
        S.xl[0] = ldb4(rx, o.x * 4u, cs); S.xh[0] = ldb4(rx, o.x * 4u + 16u, cs);
        S.xl[1] = ldb4(rx, o.y * 4u, cs); S.xh[1] = ldb4(rx, o.y * 4u + 16u, cs);
        S.xl[2] = ldb4(rx, o.z * 4u, cs); S.xh[2] = ldb4(rx, o.z * 4u + 16u, cs);
        S.xl[3] = ldb4(rx, o.w * 4u, cs); S.xh[3] = ldb4(rx, o.w * 4u + 16u, cs);
        if constexpr (PROC == LF_PRO_BNRELU) {
            const unsigned c8 = (unsigned)(kq * 8) * 4u;
            S.sc0 = ldb4(rsc, c8, cs); S.sc1 = ldb4(rsc, c8 + 16u, cs);
            S.sh0 = ldb4(rsh, c8, cs); S.sh1 = ldb4(rsh, c8 + 16u, cs);
        }
        S.ok = live ? okb : 0u;
        const int cbn = cb_ld + 1;
        const bool wrap = cbn == ncb;
        cb_ld = live ? (wrap ? 0 : cbn) : cb_ld;
        t_ld = (live && wrap) ? t_ld + 1 : t_ld;
    };
    static_assert(MT == 4, "tab_off packs 4 pixel tiles");
    static_assert(NT == 4, "group A (256 threads) stages a 64-channel weight slab");
    Raw R;
    bf16x8 xb[MT][3];
    if constexpr (DBG) tstamp[1] = __builtin_amdgcn_s_memrealtime();
    if (grp == 0) wfetch();       // W[0]
    issue(R);                     // pixels of step 0
    if (grp == 1) __syncthreads();            // B starts one phase late
    for (int step = 0; step < nsteps; ++step) {
        // ---- VALU phase: split this step's pixels, start the next loads; A publishes W[step]
        if (grp == 0) { wstore(step & 1); wfetch(); }
        if constexpr (PROC == LF_PRO_BNRELU) {
#pragma unroll
            for (int m = 0; m < MT; ++m) { R.xl[m] = max0(R.xl[m] * R.sc0 + R.sh0); R.xh[m] = max0(R.xh[m] * R.sc1 + R.sh1); }
        }
        if (__builtin_amdgcn_ballot_w64(R.ok != 15u) != 0ull) {      // wave-uniform: only tiles that touch the padding mask
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool in = (R.ok >> m) & 1u;
                f32x4 lo = R.xl[m], hi = R.xh[m];
                lo.x = in ? lo.x : 0.f; lo.y = in ? lo.y : 0.f; lo.z = in ? lo.z : 0.f; lo.w = in ? lo.w : 0.f;
                hi.x = in ? hi.x : 0.f; hi.y = in ? hi.y : 0.f; hi.z = in ? hi.z : 0.f; hi.w = in ? hi.w : 0.f;
                R.xl[m] = lo; R.xh[m] = hi;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) split3(R.xl[m], R.xh[m], xb[m][0], xb[m][1], xb[m][2]);
        issue(R);                 // next step's pixels (clamped / masked past the end) fly during the matrix phase
        // the split must be DONE before the barrier: without the scheduling fence hipcc sinks the whole VALU block
        // below s_barrier (register-only code is free to move), i.e. into this group's own matrix phase
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) asm volatile("" ::"v"(xb[m][pc]));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix phase (the partner group is in its VALU phase)
        const int cur = step & 1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wl[cur][0][kq][n * 16 + pl]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, wl[cur][1][kq][n * 16 + pl]);
            const bf16x8 wo = __builtin_bit_cast(bf16x8, wl[cur][2][kq][n * 16 + pl]);
            // smallest terms first; consecutive MFMAs of one term go to four different accumulators
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo, xb[m][2], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo, xb[m][1], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xb[m][2], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo, xb[m][0], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xb[m][2], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xb[m][1], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xb[m][0], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xb[m][1], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xb[m][0], acc[n][m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __syncthreads();            // A pairs B's extra first barrier (B's last matrix phase)
    if constexpr (DBG) {
        asm volatile("" ::"v"(acc[0][0][0]));
        tstamp[2] = __builtin_amdgcn_s_memrealtime();
    }
#undef LF_EPI_PIVOT
#define LF_EPI_PIVOT (EPIC >= 0)
    LF_TAPGEMM_EPILOGUE
#undef LF_EPI_PIVOT
#define LF_EPI_PIVOT true
    if constexpr (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.dbg) {
            unsigned long long* d = a.dbg + ((unsigned long long)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 8;
            d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3];
        }
    }
}
#undef LF_EPI_GROUPS
#define LF_EPI_GROUPS 1

// Lean variant for 16-output-channel launches (the 128x256 stage, ~3 % of the FLOPs): these are HBM-bound
// (0.75*C = 12 FLOP/B), so the goal is bytes in flight, not MFMA issue: no operand ring, few registers,
// many waves per SIMD; the compiler is free to hoist the next step's loads.
// A wave of these launches is 48 MFMAs between a prologue and an epilogue of VALU work, 16 waves per SIMD in sequence: the
// index arithmetic IS the kernel.  PROC / EPIC compiled in (EPIC = -1: run-time flags); with Wl % 64 == 0 a wave's 64 pixels lie
// in one image row, so (image, row, first column) are wave-uniform: two divisions instead of eight, scalar row terms.
template <int NT, int PROC = -1, int EPIC = -1>
__global__ __launch_bounds__(256, 4) void tapgemm_lean_kernel(const LfTapGeom g, const LfTapArgs a, const int pro_rt, const int epi_rt) {
    constexpr bool S16 = false, HOISTV = EPIC >= 0;
    const int pro = PROC >= 0 ? PROC : pro_rt, epi = EPIC >= 0 ? EPIC : epi_rt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, kq = lane >> 4;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl);
    const int cob = blockIdx.y * NT * 16;
    unsigned bx = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const unsigned tile0 = (bx * WG_WAVES + __builtin_amdgcn_readfirstlane(wave)) * (MT * 16);
    int pn[MT], pi[MT], pj[MT];
    bool pv[MT];
    const bool onerow = (g.Wl & 63) == 0;
    if (onerow) {
        const unsigned q = tile0 < npix ? tile0 : 0u;                 // wave-uniform
        const unsigned r = q / (unsigned)g.Wl;
        const int j0 = (int)(q - r * (unsigned)g.Wl), n0 = (int)(r / (unsigned)g.Hl), i0 = (int)(r - (unsigned)n0 * (unsigned)g.Hl);
#pragma unroll
        for (int m = 0; m < MT; ++m) { pv[m] = tile0 < npix; pn[m] = n0; pi[m] = i0; pj[m] = j0 + m * 16 + pl; }
    } else {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const unsigned p = tile0 + m * 16 + pl;
        pv[m] = p < npix;
        const unsigned q = pv[m] ? p : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj[m] = (int)(q - r * (unsigned)g.Wl);
        pn[m] = (int)(r / (unsigned)g.Hl);
        pi[m] = (int)(r - (unsigned)pn[m] * (unsigned)g.Hl);
    }
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = zero4();
    f32x4 seg[NT][MT];                 // running chain of the general path (see below)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) seg[n][m] = zero4();
    const int ncg = g.Cs >> 4;
    const float* wp = a.wp + (long)(kq * g.Cd + cob + pl) * 4;
    // x: num_records = the tensor, so that LF_OOB reads as 0.0f (padding without masks when there is no prologue)
    const __amdgpu_buffer_rsrc_t rwl = make_rsrc(a.wp, 0xffffffffu),
                                 rxl = make_rsrc(a.src, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 4, (long)LF_OOB));
    const unsigned wlane = (unsigned)(kq * g.Cd + cob + pl) * 16u;
    if (g.Cs == 16 && g.ntaps == 3) {
        // the 16 -> 16 channel 1-D convs (24 launches per step): all three taps' operands are requested before the first
        // MFMA -- one memory round trip per wave instead of three (these launches are HBM-bound, a wave is ~0.2 us of MFMAs)
        f32x4 w3[3][NT], x3[3][MT];
        bool in3[3][MT];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int dh = g.tdh[t], dw = g.tdw[t];
#pragma unroll
            for (int n = 0; n < NT; ++n) w3[t][n] = ldb4(rwl, wlane + n * 256, (unsigned)(t * g.Cd * 64));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int sy = pi[m] * g.ssh + dh, sx = pj[m] * g.ssw + dw;
                in3[t][m] = pv[m] && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
                const int syc = min(max(sy, 0), g.Hs - 1), sxc = min(max(sx, 0), g.Ws - 1);
                const unsigned off = (unsigned)(((pn[m] * g.Hs + syc) * g.Ws + sxc) * g.s_pix + g.s_choff + kq * 4) * 4u;
                x3[t][m] = ldb4(rxl, (PROC == 0 && !in3[t][m]) ? LF_OOB : off, 0u);
            }
        }
        f32x4 sc = zero4(), sh = zero4();
        if (pro == LF_PRO_BNRELU) { sc = ldg4(a.pro_sc + kq * 4); sh = ldg4(a.pro_sh + kq * 4); }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if constexpr (PROC != 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 v = x3[t][m];
                    if (pro == LF_PRO_BNRELU) v = max0(v * sc + sh);
                    v.x = in3[t][m] ? v.x : 0.f; v.y = in3[t][m] ? v.y : 0.f; v.z = in3[t][m] ? v.z : 0.f; v.w = in3[t][m] ? v.w : 0.f;
                    x3[t][m] = v;
                }
            }
            // one fma chain per TAP (16 products), the three summed afterwards: the summation-error rule of tapgemm_kernel
            // (segments of <= 32 products), here for the layers next to the logits
            f32x4 part[NT][MT];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        part[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[t][n][s4], x3[t][m][s4], s4 ? part[n][m] : zero4(), 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = t ? acc[n][m] + part[n][m] : part[n][m];
        }
    } else
    for (int t = 0; t < g.ntaps; ++t) {
        const int dh = g.tdh[t], dw = g.tdw[t];
        unsigned xo[MT];
        bool in[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int sy = pi[m] * g.ssh + dh, sx = pj[m] * g.ssw + dw;
            in[m] = pv[m] && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
            const int syc = min(max(sy, 0), g.Hs - 1), sxc = min(max(sx, 0), g.Ws - 1);
            xo[m] = (unsigned)(((pn[m] * g.Hs + syc) * g.Ws + sxc) * g.s_pix + g.s_choff + kq * 4);
        }
        for (int cg = 0; cg < ncg; ++cg) {
            f32x4 w[NT], x[MT];
#pragma unroll
            for (int n = 0; n < NT; ++n) w[n] = ldg4(wp + n * 64);
            wp += (long)g.Cd * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m) x[m] = ldg4(a.src + xo[m] + cg * 16);
            if (pro == LF_PRO_BNRELU) {
                const f32x4 sc = ldg4(a.pro_sc + cg * 16 + kq * 4), sh = ldg4(a.pro_sh + cg * 16 + kq * 4);
#pragma unroll
                for (int m = 0; m < MT; ++m) x[m] = max0(x[m] * sc + sh);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                x[m].x = in[m] ? x[m].x : 0.f; x[m].y = in[m] ? x[m].y : 0.f;
                x[m].z = in[m] ? x[m].z : 0.f; x[m].w = in[m] ? x[m].w : 0.f;
            }
            // (a chain restarts every two channel groups = 32 products and is added into acc: tapgemm_kernel's rule)
            if ((cg & 1) == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m) { acc[n][m] += seg[n][m]; seg[n][m] = zero4(); }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        seg[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][s], x[m][s], seg[n][m], 0, 0, 0);
        }
    }
    if (!(g.Cs == 16 && g.ntaps == 3)) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] += seg[n][m];
    }
    LF_TAPGEMM_EPILOGUE
}

int g_split_any_size = 0;      // kernel-level tests only: let the split kernel take launches below its shipped size rule
int g_bf16_lds = 4;            // tools / A-B runs only: 4 = wave-private (64 ch) / whole-line (128 ch) + 16-channel kernels where they apply, else the ring (shipped);
                               // 3 = the whole-line kernel at 64 channels too (round 5's routing); 2 = the ring
                               // for every launch it takes; 0 = the streaming bf16 kernel only

int pick_nt(int Cd) {
    const int tiles = Cd / 16;
    if (tiles % 4 == 0) return 4;
    if (tiles % 3 == 0) return 3;
    if (tiles % 2 == 0) return 2;
    return 1;
}

}  // namespace

void lf_tapgemm_set_split_any_size(int v) { g_split_any_size = v; }
void lf_tapgemm_set_bf16_lds(int v) { g_bf16_lds = v; }

// launches the split kernel takes: whole 32-channel K-steps, 64-channel output slabs (NT = 4), whole 512-pixel
// workgroups (its two 4-wave groups each own one 256-pixel statistics row), 16-byte aligned pixels
bool lf_tapgemm_split_ok(const LfTapGeom& g) {
    const long npix = (long)g.N * g.Hl * g.Wl;
    // its 512-pixel workgroups run one per CU: a launch that cannot put one on (most of) the 256 CUs is faster on the fp32
    // cores with their 256-pixel workgroups, two per CU (batch 16: 1412 vs 1484 images/s before this rule)
    const long wgs = npix / (2 * PIX_PER_WG) * (g.Cd / 64);
    return g.Cs % 32 == 0 && g.Cd % 64 == 0 && g.s_pix % 4 == 0 && g.s_choff % 4 == 0 && npix % (2 * PIX_PER_WG) == 0 &&
           (wgs >= 192 || g_split_any_size);
}

int lf_tapgemm_stat_rows(const LfTapGeom& g) {
    const long npix = (long)g.N * g.Hl * g.Wl;
    return lf_cdiv(npix, PIX_PER_WG);
}
int lf_tapgemm_stat_rows_for(const LfTapGeom& g, const LfTapArgs& a) {
    (void)a;
    return lf_cdiv((long)g.N * g.Hl * g.Wl, PIX_PER_WG);
}

namespace {
// Per-(device, kernel) launch state shared by the host threads of a process (one process may drive several GPUs): the occupancy
// cache below and the "dynamic LDS beyond 64 KB allowed" attribute of the LDS-ring kernels.
std::mutex g_launch_state_mutex;
int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
// Workgroups of `kernel` the whole chip holds at once (occupancy x CUs), cached per (device, kernel, dynamic LDS bytes).
template <typename K>
int resident_workgroups(K kernel, size_t lds) {
    static std::map<std::tuple<int, const void*, size_t>, int> cache;
    const int dev = current_device();
    const std::tuple<int, const void*, size_t> key(dev, reinterpret_cast<const void*>(kernel), lds);
    std::lock_guard<std::mutex> lock(g_launch_state_mutex);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int cus = 0, nb = 0;
    if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, lds) != hipSuccess || nb < 1 || cus < 1)
        return cache[key] = 1 << 30;                 // unknown: one tile per workgroup
    return cache[key] = nb * cus;
}
// hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel); false when the runtime refuses (the caller launches a
// kernel that needs less LDS instead of failing at launch)
bool allow_big_lds(const void* kernel, int bytes) {
    static std::map<std::pair<int, const void*>, bool> done;
    const std::pair<int, const void*> key(current_device(), kernel);
    std::lock_guard<std::mutex> lock(g_launch_state_mutex);
    auto it = done.find(key);
    if (it != done.end()) return it->second;
    return done[key] = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}
// tapgemm_kernel walks its pixel tiles with stride gridDim.x: launch no more workgroups than are resident at once (a multiple
// of 8 per XCD dealing), so that the tiles beyond the first round are spread one per CU by construction
// hipExtAnyOrderLaunch for the next launch_tapgemm (set by lf_tapgemm_launch_unordered): the dispatch packet carries no barrier
// bit, so the kernel may start while the PREVIOUS packet of the stream is still running -- its workgroups fill the slots the
// previous kernel's last workgroups leave (the caller guarantees the two are independent).
thread_local unsigned g_launch_flags = 0;
template <typename K>
void launch_tapgemm(K kernel, dim3 grid, size_t lds, hipStream_t st, const LfTapGeom& g, const LfTapArgs& a, int pro, int epi) {
    const unsigned res = (unsigned)resident_workgroups(kernel, lds) / grid.y;
    if (grid.x > res && res >= 8) grid.x = res & ~7u;
    if (g_launch_flags) hipExtLaunchKernelGGL(kernel, grid, dim3(256), (unsigned)lds, st, nullptr, nullptr, g_launch_flags, g, a, pro, epi);
    else hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, g, a, pro, epi);
}
// tapgemm_bf16_ring_kernel: persistent, at most the resident workgroups (a multiple of 8: XCD-contiguous item ranges)
template <int EPIV, bool DBGV = false>
void launch_bf16_ring(unsigned nitems, hipStream_t st, const LfTapGeom& g, const LfTapArgs& a, int pro, int epi) {
    auto kern = tapgemm_bf16_ring_kernel<EPIV, DBGV>;
    const size_t ring_lds = (size_t)LB_STAGES * LB_STAGE_BYTES;
    unsigned gx = nitems;
    const unsigned res = (unsigned)resident_workgroups(kern, ring_lds);
    if (gx > res && res >= 8) gx = res & ~7u;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), ring_lds, st, g, a, pro, epi);
}
// false: the runtime refused the kernel's LDS budget on this device (nothing launched: the caller takes the ring / streaming kernel)
template <int CBV, int EPIV, int PROV = 0>
bool launch_bf16_wl(unsigned nitems, hipStream_t st, const LfTapGeom& g, const LfTapArgs& a, int pro, int epi) {
    auto kern = tapgemm_bf16_wl_kernel<CBV, EPIV, PROV>;
    const size_t lds = WlCfg<CBV, EPIV, PROV>::LDS;
    if (!allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 4096)) return false;
    unsigned gx = nitems;
    const unsigned res = (unsigned)resident_workgroups(kern, lds);
    if (gx > res && res >= 8) gx = res & ~7u;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, st, g, a, pro, epi);
    return true;
}
// tapgemm_bf16_wv_kernel (64 channels): same contract
template <int EPIV>
bool launch_bf16_wv(unsigned ntiles, hipStream_t st, const LfTapGeom& g, const LfTapArgs& a, int pro, int epi) {
    auto kern = tapgemm_bf16_wv_kernel<EPIV>;
    const size_t lds = WvCfg<EPIV>::LDS;
    if (!allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 4096)) return false;
    unsigned gx = ntiles;
    const unsigned res = (unsigned)resident_workgroups(kern, lds);
    if (gx > res && res >= 8) gx = res & ~7u;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, st, g, a, pro, epi);
    return true;
}
}  // namespace

int lf_tapgemm_launch_unordered(const LfTapGeom& g, const LfTapArgs& a, int pro, int epi, hipStream_t st) {
    g_launch_flags = hipExtAnyOrderLaunch;
    const int rc = lf_tapgemm_launch(g, a, pro, epi, st);
    g_launch_flags = 0;
    return rc;
}

int lf_tapgemm_launch(const LfTapGeom& g, const LfTapArgs& a, int pro, int epi, hipStream_t st) {
    LF_REQUIRE(g.Cs % 16 == 0 && g.Cd % 16 == 0, "tapgemm: channels must be multiples of 16 (Cs=%d Cd=%d)", g.Cs, g.Cd);
    LF_REQUIRE(g.s_pix % 4 == 0 && g.s_choff % 4 == 0 && g.d_pix % 4 == 0 && g.d_choff % 4 == 0, "tapgemm: unaligned channel layout");
    LF_REQUIRE(g.ntaps >= 1 && g.ntaps <= LF_MAX_TAPS, "tapgemm: bad tap count %d", g.ntaps);
    const long npix = (long)g.N * g.Hl * g.Wl;
    LF_REQUIRE(npix < (1L << 30), "tapgemm: too many pixels (%ld)", npix);
    const int nt = pick_nt(g.Cd);
    dim3 grid(lf_cdiv(npix, PIX_PER_WG), g.Cd / (16 * nt));
    LF_REQUIRE((long)g.N * g.Hs * g.Ws * g.s_pix * 4 < (long)LF_OOB, "tapgemm: source tensor too large for 32-bit byte offsets");
    LF_REQUIRE((long)g.N * g.Hd * g.Wd * g.d_pix * 4 < (long)LF_OOB, "tapgemm: destination tensor too large for 32-bit byte offsets");
#define LF_TG(NTV)                                                                                                       \
    do {                                                                                                                 \
        if (pro == LF_PRO_BNRELU) launch_tapgemm(tapgemm_kernel<NTV, 1>, grid, tap_lds, st, g, a, pro, epi);  \
        else launch_tapgemm(tapgemm_kernel<NTV, 0>, grid, tap_lds, st, g, a, pro, epi);                       \
    } while (0)
#define LF_TG4(PROV, EPIV)                                                                                               \
    do {                                                                                                                 \
        if ((g.Wl & 63) == 0) launch_tapgemm(tapgemm_kernel<4, PROV, EPIV, true>, grid, tap_lds, st, g, a, pro, epi);  \
        else launch_tapgemm(tapgemm_kernel<4, PROV, EPIV, false>, grid, tap_lds, st, g, a, pro, epi);    \
    } while (0)
    const size_t tap_lds = LF_TAP_LDS_PER_TAP * g.ntaps;
    // the compiled-in data-gradient epilogues (mask / residual / BN-backward sums) carry no bias vector (NOBIAS): a launch
    // that combines one of them with a bias takes the run-time-flag kernel
    const int epis = ((epi & (LF_EPI_MASK | LF_EPI_ADD | LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) && a.bias) ? -2 : epi;
    LF_REQUIRE(!a.s16 || a.wp16, "tapgemm: bf16 tensors need the bf16 matrix-core kernel (wp16)");
    if (a.split && a.wp48 && !a.wp16 && lf_tapgemm_split_ok(g)) {
        LF_REQUIRE(a.split == 9, "tapgemm: split must be 9 (got %d; the 6-term form was removed in round 6)", a.split);
        LfTapArgs b = a;
        b.wp16 = a.wp48;
        const dim3 grid2((unsigned)(npix / (2 * PIX_PER_WG)), g.Cd / 64);     // 512-pixel workgroups (two 4-wave groups)
        if (a.dbg) {       // phase stamps (tools/kbench.py --phases): the plain conv only
            LF_REQUIRE(pro == LF_PRO_NONE && epi == 0, "tapgemm: phase stamps are compiled into the plain convolution only");
            hipLaunchKernelGGL((tapgemm_split_kernel<4, 0, 0, true>), grid2, dim3(512), 0, st, g, b, pro, epi);
        } else {
#define LF_TS9(PROV, EPIV) hipLaunchKernelGGL((tapgemm_split_kernel<4, PROV, EPIV>), grid2, dim3(512), 0, st, g, b, pro, epi)
            if (pro == LF_PRO_BNRELU && epi == LF_EPI_RELU) LF_TS9(1, LF_EPI_RELU);
            else if (pro == LF_PRO_BNRELU) LF_TS9(1, -1);
            else switch (epis) {
                case 0: LF_TS9(0, 0); break;
                case LF_EPI_RELU: LF_TS9(0, LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TS9(0, LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TS9(0, LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TS9(0, LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TS9(0, LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TS9(0, LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TS9(0, LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TS9(0, -1); break;
            }
#undef LF_TS9
        }
        LF_CHECK_LAUNCH("tapgemm_split");
        return 0;
    }
    LF_REQUIRE(!a.s16 || (g.s_pix % 8 == 0 && g.s_choff % 8 == 0), "tapgemm: bf16 source layout must be 16-byte aligned per pixel");
    if (a.wp16) {
#define LF_TG16(NTV)                                                                                                     \
    do {                                                                                                                 \
        if (pro == LF_PRO_BNRELU) hipLaunchKernelGGL((tapgemm_bf16_kernel<NTV, 1>), grid, dim3(256), 0, st, g, a, pro, epi); \
        else hipLaunchKernelGGL((tapgemm_bf16_kernel<NTV, 0>), grid, dim3(256), 0, st, g, a, pro, epi);                  \
    } while (0)
        LF_REQUIRE(a.s16, "tapgemm: the bf16 matrix-core kernels take bf16 tensors (s16); bf16 operands on fp32 tensors were removed in round 6");
        LF_REQUIRE(g.Cs >= 8 && g.s_pix >= g.s_choff + 8, "tapgemm bf16: needs at least 8 source channels");
#define LF_TG16F(EPIV) hipLaunchKernelGGL((tapgemm_bf16_kernel<4, 0, EPIV, true>), grid, dim3(256), 0, st, g, a, pro, epi)
        // the 16 -> 16 channel 3-tap convolutions on bf16 tensors: tapgemm_bf16_lean_kernel
        if (a.s16 && g_bf16_lds >= 3 && !a.dbg && g.Cs == 16 && g.Cd == 16 && g.ntaps == 3 && g.Wl % 16 == 0 && g.s_pix % 8 == 0 && g.s_choff % 8 == 0 &&
            (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB) {
#define LF_TGL(EPIV) do { if (pro == LF_PRO_BNRELU) hipLaunchKernelGGL((tapgemm_bf16_lean_kernel<1, EPIV>), grid, dim3(256), 0, st, g, a, pro, epi); \
                          else hipLaunchKernelGGL((tapgemm_bf16_lean_kernel<0, EPIV>), grid, dim3(256), 0, st, g, a, pro, epi); } while (0)
            switch (epis) {
                case 0: LF_TGL(0); break;
                case LF_EPI_RELU: LF_TGL(LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TGL(LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TGL(LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TGL(LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGL(LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGL(LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TGL(LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TGL(-1); break;
            }
#undef LF_TGL
            LF_CHECK_LAUNCH("tapgemm_bf16_lean");
            return 0;
        }
        const bool fast16 = nt == 4 && a.s16 && pro != LF_PRO_BNRELU && g.Cs % 32 == 0 &&
                            (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB;
        // Which kernel (g_bf16_lds = 4, the shipped setting): the 3-tap convolutions of non_bottleneck_1d at 64 / 128 channels ->
        // tapgemm_bf16_wl_kernel (whole lines); every other prologue-free launch at 64-channel output slabs whose 16-pixel groups lie
        // in one image row (the 9-tap stride-2 convolution, the transposed-convolution phases, their gradients) ->
        // tapgemm_bf16_ring_kernel; the rest (operand prologue, ragged widths) -> the streaming tapgemm_bf16_kernel
        // the block's third convolution (BN+ReLU operand prologue), at 128 channels: 83 -> 60 us per launch at config 3's size; at 64
        // channels the in-LDS transform is as long as the whole step (88 -> 91 us): those stay on the streaming kernel
        const bool fast16p = nt == 4 && a.s16 && pro == LF_PRO_BNRELU && epi == LF_EPI_RELU && g.Cs == 128 &&
                             (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB;
        const bool line3 = g_bf16_lds >= 3 && !a.dbg && g.ntaps == 3 && g.Cs == g.Cd && (g.Cd == 64 || g.Cd == 128) && g.Wl % 16 == 0 &&
                           g.ssh == 1 && g.ssw == 1 && g.dsh == 1 && g.dsw == 1 && g.dah == 0 && g.daw == 0 && g.Hs == g.Hl && g.Ws == g.Wl &&
                           g.Hd == g.Hl && g.Wd == g.Wl && g.s_pix % 8 == 0 && g.s_choff % 8 == 0 && g.d_pix % 8 == 0 && g.d_choff % 8 == 0 &&
                           (long)g.N * g.Hd * g.Wd * g.d_pix * 2 < (long)LF_OOB;
        // 64 channels (g_bf16_lds = 4, shipped): tapgemm_bf16_wv_kernel -- a wave owns its pixels and all 64 output channels; 3 = the
        // whole-line kernel for both channel counts (round 5's routing, kept for A/B runs: tools/bf16_ab.py)
        const bool wv = line3 && g_bf16_lds == 4 && g.Cd == 64 && fast16;
        const bool wl = line3 && (fast16 || fast16p);      // (also what the wave-private kernel declines)
        if (wv) {
            const unsigned ntiles = (unsigned)lf_cdiv(npix, PIX_PER_WG);
            bool launched = false;
#define LF_TGV(EPIV) do { launched = launch_bf16_wv<EPIV>(ntiles, st, g, a, pro, epi); } while (0)
            switch (epis) {
                case 0: LF_TGV(0); break;
                case LF_EPI_RELU: LF_TGV(LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TGV(LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TGV(LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TGV(LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGV(LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                // (ADD + MASK + BN-backward sums: two staged operand tiles leave room for ONE workgroup per CU here -- 123.6 us in config
                // 3's step against 115.8 us on the whole-line kernel, which takes these launches: `launched` stays false)
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TGV(LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TGV(-1); break;
            }
#undef LF_TGV
            if (launched) {
                LF_CHECK_LAUNCH("tapgemm_bf16_wv");
                return 0;
            }
        }
        if (wl && fast16p) {        // (128 channels only)
            if (launch_bf16_wl<4, LF_EPI_RELU, 1>((unsigned)lf_cdiv(npix, PIX_PER_WG), st, g, a, pro, epi)) {
                LF_CHECK_LAUNCH("tapgemm_bf16_wl (prologue)");
                return 0;
            }
        } else if (wl) {
            const unsigned nitems = (unsigned)lf_cdiv(npix, PIX_PER_WG);
            bool launched = false;
#define LF_TGW(EPIV) do { launched = g.Cs == 128 ? launch_bf16_wl<4, EPIV>(nitems, st, g, a, pro, epi) : launch_bf16_wl<2, EPIV>(nitems, st, g, a, pro, epi); } while (0)
            switch (epis) {
                case 0: LF_TGW(0); break;
                case LF_EPI_RELU: LF_TGW(LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TGW(LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TGW(LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TGW(LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGW(LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGW(LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TGW(LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TGW(-1); break;
            }
#undef LF_TGW
            if (launched) {
                LF_CHECK_LAUNCH("tapgemm_bf16_wl");
                return 0;
            }
        }
        const bool ring = fast16 && g_bf16_lds >= 2 && g.Wl % 16 == 0 && g.Cd % 64 == 0;
        const unsigned nitems = (unsigned)(lf_cdiv(npix, PIX_PER_WG) * (g.Cd / 64));
        if (ring && a.dbg) {        // stamps (tools/kbench.py --phases16): the plain convolution only
            LF_REQUIRE(epis == 0, "tapgemm bf16 ring: stamps are compiled into the plain convolution only");
            launch_bf16_ring<0, true>(nitems, st, g, a, pro, epi);
        } else
        if (ring) {
#define LF_TGR(EPIV) launch_bf16_ring<EPIV>(nitems, st, g, a, pro, epi)
            switch (epis) {
                case 0: LF_TGR(0); break;
                case LF_EPI_RELU: LF_TGR(LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TGR(LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TGR(LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TGR(LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGR(LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TGR(LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TGR(LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TGR(-1); break;
            }
#undef LF_TGR
        } else if (fast16) {       // the bf16-tensor launches of the network at 64 output channels per workgroup
            switch (epis) {
                case 0: LF_TG16F(0); break;
                case LF_EPI_RELU: LF_TG16F(LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TG16F(LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TG16F(LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TG16F(LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TG16F(LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TG16F(LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TG16F(LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TG16F(-1); break;
            }
        } else if (nt == 4 && a.s16 && pro != LF_PRO_BNRELU && g.Cs == 16 && (epis == (LF_EPI_MASK | LF_EPI_STATS_XHAT) || epis == 0) &&
                   (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB) {
            // 16 source channels into a 64-channel slab (the data gradient of UpsamplerBlock(64, 16): 9 taps, stride 2, with the previous
            // layer's mask + BN-backward sums; 204 us per launch at config 3 on the run-time-flag form): ONE 32-channel step per tap on the
            // compiled-in form.  Lanes kq = 2, 3 read the 16 elements BEHIND the pixel's 16 channels -- the next pixel's (or, past the
            // tensor's end, the buffer bound's zeros): finite values against the zero-padded half of the packed weights
            if (epis == 0) LF_TG16F(0); else LF_TG16F(LF_EPI_MASK | LF_EPI_STATS_XHAT);
        } else if (nt == 3 && a.s16 && pro != LF_PRO_BNRELU && g.Cs == 16 && (epis == LF_EPI_STATS_SQ || epis == 0) &&
                   (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB) {
            // ... and the 16 -> 48 channel convolution of DownsamplerBlock(16, 64) (9 taps, stride 2, BN forward sums; 112 us on the run-time-flag form)
            if (epis == 0) hipLaunchKernelGGL((tapgemm_bf16_kernel<3, 0, 0, true>), grid, dim3(256), 0, st, g, a, pro, epi);
            else hipLaunchKernelGGL((tapgemm_bf16_kernel<3, 0, LF_EPI_STATS_SQ, true>), grid, dim3(256), 0, st, g, a, pro, epi);
        } else if (nt == 4 && a.s16 && pro == LF_PRO_BNRELU && epi == LF_EPI_RELU) {
            hipLaunchKernelGGL((tapgemm_bf16_kernel<4, 1, LF_EPI_RELU, false>), grid, dim3(256), 0, st, g, a, pro, epi);
        } else if (nt == 1 && a.s16 && pro != LF_PRO_BNRELU && ((g.Cs + 31) / 32 * 32 + g.s_choff <= g.s_pix) && (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB &&
                   (epis == 0 || epis == LF_EPI_STATS_SQ || epis == LF_EPI_ADD)) {
            // 16 output channels from whole 32-channel steps (round 6: the sub-pixel phases of UpsamplerBlock(64, 16) and the data gradient
            // of DownsamplerBlock(16, 64)'s convolution: 8 launches per step on the run-time-flag form before): padding as out-of-range
            // offsets, compiled-in epilogue.  A partial last step (48 source channels) reads the pixel's next channels -- they exist:
            // the condition above -- against zero-padded weights
            if (epis == 0) hipLaunchKernelGGL((tapgemm_bf16_kernel<1, 0, 0, true>), grid, dim3(256), 0, st, g, a, pro, epi);
            else if (epis == LF_EPI_STATS_SQ) hipLaunchKernelGGL((tapgemm_bf16_kernel<1, 0, LF_EPI_STATS_SQ, true>), grid, dim3(256), 0, st, g, a, pro, epi);
            else hipLaunchKernelGGL((tapgemm_bf16_kernel<1, 0, LF_EPI_ADD, true>), grid, dim3(256), 0, st, g, a, pro, epi);
        } else
        switch (nt) {
            case 4: LF_TG16(4); break;
            case 3: LF_TG16(3); break;
            case 2: LF_TG16(2); break;
            default: LF_TG16(1); break;
        }
#undef LF_TG16
#undef LF_TG16F
        LF_CHECK_LAUNCH("tapgemm_bf16");
        return 0;
    }
    if (a.dbg) {           // phase stamps (tools/kbench.py --phases): the plain 64-channel-slab convolution only
        LF_REQUIRE(nt == 4 && pro == LF_PRO_NONE && epi == 0 && !a.wp16, "tapgemm: phase stamps are compiled into the plain fp32 convolution only");
        if ((g.Wl & 63) == 0) hipLaunchKernelGGL((tapgemm_kernel<4, 0, 0, true, true>), grid, dim3(256), tap_lds, st, g, a, pro, epi);
        else hipLaunchKernelGGL((tapgemm_kernel<4, 0, 0, false, true>), grid, dim3(256), tap_lds, st, g, a, pro, epi);
        LF_CHECK_LAUNCH("tapgemm (stamps)");
        return 0;
    }
    switch (nt) {
        case 4:
            if (pro == LF_PRO_BNRELU && epi == LF_EPI_RELU) LF_TG4(1, LF_EPI_RELU);
            else if (pro == LF_PRO_BNRELU) LF_TG4(1, -1);
            else switch (epis) {
                case 0: LF_TG4(0, 0); break;
                case LF_EPI_RELU: LF_TG4(0, LF_EPI_RELU); break;
                case LF_EPI_MASK: LF_TG4(0, LF_EPI_MASK); break;
                case LF_EPI_ADD: LF_TG4(0, LF_EPI_ADD); break;
                case LF_EPI_STATS_SQ: LF_TG4(0, LF_EPI_STATS_SQ); break;
                case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TG4(0, LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_TG4(0, LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_TG4(0, LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                default: LF_TG4(0, -1); break;
            }
            break;
        case 3: LF_TG(3); break;
        case 2: LF_TG(2); break;
        default:
            {
#define LF_LEAN(PROV, EPIV) hipLaunchKernelGGL((tapgemm_lean_kernel<1, PROV, EPIV>), grid, dim3(256), 0, st, g, a, pro, epi)
                if (pro == LF_PRO_BNRELU && epi == LF_EPI_RELU) LF_LEAN(1, LF_EPI_RELU);
                else if (pro == LF_PRO_BNRELU) LF_LEAN(1, -1);
                else switch (epis) {
                    case 0: LF_LEAN(0, 0); break;
                    case LF_EPI_RELU: LF_LEAN(0, LF_EPI_RELU); break;
                    case LF_EPI_MASK: LF_LEAN(0, LF_EPI_MASK); break;
                    case LF_EPI_ADD: LF_LEAN(0, LF_EPI_ADD); break;
                    case LF_EPI_STATS_SQ: LF_LEAN(0, LF_EPI_STATS_SQ); break;
                    case LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_LEAN(0, LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                    case LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT: LF_LEAN(0, LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT); break;
                    case LF_EPI_MASKBN | LF_EPI_STATS_XHAT: LF_LEAN(0, LF_EPI_MASKBN | LF_EPI_STATS_XHAT); break;
                    default: LF_LEAN(0, -1); break;
                }
#undef LF_LEAN
            }
            break;
    }
#undef LF_TG
#undef LF_TG4
    LF_CHECK_LAUNCH("tapgemm");
    return 0;
}

// ---------------------------------------------------------------------------------------
// weight gradient: dW[t][ci][co] = sum_p pro(X[spix(p,t)][ci]) * G[dpix(p)][co]
// rows = x-channels, cols = g-channels, k = 4 pixels per MFMA (k-slot kq <-> pixel p0+kq).
// Vector mode (channels % 64 == 0): a lane's float4 = channels 4*pl..4*pl+3 of its pixel feeds
// FOUR MFMA tiles (tile r = channels {4*i + r}); so 2 dwordx4 loads feed 16 MFMAs.
// ---------------------------------------------------------------------------------------
namespace {

// storage-typed loads of the weight-gradient kernels (S16: x and g hold bf16 elements, widened exactly to fp32)
// (buffer-addressed, element offsets: see ldb4)
template <bool S16>
__device__ __forceinline__ f32x4 wg_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) { return epi_ld<S16>(r, off); }
template <bool S16>
__device__ __forceinline__ float wg_ld1(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff = 0u) {
    if constexpr (S16) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, (int)(off * 2u), (int)(soff * 2u), 0) << 16);
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(off * 4u), (int)(soff * 4u), 0));
}
__device__ __forceinline__ uint2 wg_ldraw(__amdgpu_buffer_rsrc_t r, unsigned off) {      // 4 bf16 elements, raw
    return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)(off * 2u), 0, 0));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// BFM (bf16 tensors, 64-channel blocks on both sides): the four K=4 fp32 MFMAs of an iteration's four pixel
// sub-steps become ONE v_mfma_f32_16x16x16_bf16 per tile -- a lane's four pixels (p+kq, +4, +8, +12) are exactly its
// four k of the K=16 step -- with the operands assembled from the raw 8-byte loads by v_perm_b32 (no widening).
// PROT: 0 / 1 = the BN+ReLU prologue flag compiled in (the fp32 64-channel-block launches of the network), -1 = run-time
// (Round 6 built a PAIRED form -- 8-wave workgroups running the two jobs (tap, x-block, g-block 0 / 1) that read the same X stream side
// by side, one barrier per loop trip so that the partner's load finds the line in the CU's L1 -- to test whether the job form's
// re-reads bound the kernel: FETCH_SIZE fell 25 % (283 vs 376 MB per launch) and the launch did NOT get faster (64.2 vs 63.0 us, the
// step +0.13 ms); neither do operands from HBM instead of the Infinity Cache cost anything (profiles/r6_wgrad_traffic.txt).  The kernel
// is bound by issue / the matrix pipes; the form was removed (git: commit 45fe98c has it).)
template <bool XV, bool GV, int XT, int GT, int U, bool S16, bool BFM = false, int PROT = -1, bool DBG = false>
__global__ __launch_bounds__(256, 2) void tapwgrad_kernel(const LfTapGeom g, const LfWgradArgs a, const int pro_rt,
                                                      const long pps, const int write_bias, const int gxs) {
    constexpr int XTiles = XV ? 4 : XT, GTiles = GV ? 4 : GT;
    const int pro = PROT >= 0 ? (PROT ? LF_PRO_BNRELU : LF_PRO_NONE) : pro_rt;
    constexpr int XB = XTiles * 16, GB = GTiles * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, kq = lane >> 4;
    unsigned long long tstamp[4] = {0ull, 0ull, 0ull, 0ull};
    if constexpr (DBG) tstamp[0] = __builtin_amdgcn_s_memrealtime();
    // 1-D grid of gx * ntaps * (channel-block pairs) workgroups.  Workgroup L runs on XCD L % 8 (observed): each
    // XCD gets a contiguous run of the (pixel-split, channel-block, tap) order with the tap fastest, so all jobs of
    // one pixel range -- which read the same G rows and overlapping X rows -- follow each other through ONE L2
    // instead of streaming both tensors from HBM once per tap (FETCH_SIZE 177 MB -> see profiles/).
    const int ncob = g.Cd / GB;
    unsigned ord = blockIdx.x;
    if ((gridDim.x & 7u) == 0) ord = (ord & 7u) * (gridDim.x >> 3) + (ord >> 3);
    const unsigned nz = (unsigned)((g.Cs / XB) * ncob);
    const int t = (int)(ord % (unsigned)g.ntaps);
    const int bz = (int)((ord / (unsigned)g.ntaps) % nz);                 // channel-block pair: next fastest
    const unsigned bxs = ord / ((unsigned)g.ntaps * nz);                  // pixel-split index: slowest
    (void)gxs;
    const int cib = bz / ncob, cob = bz % ncob;
    // Loop state is WAVE-UNIFORM (scalar registers): a wave walks its pixel range in groups of 4U consecutive pixels of one
    // image row; lane (pl, kq) takes pixel kq + 4u of the group.  A VALU instruction issued beside the partner wave's MFMA
    // stream costs ~13 cycles, so the per-lane address / mask arithmetic of the first version (~150 instructions per 64 MFMAs)
    // was as long as the matrix work itself.
    const long npix = (long)g.N * g.Hl * g.Wl;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const long sub = (long)bxs * WG_WAVES + wave_u;
    const long p_begin = sub * pps;
    long p_end = p_begin + pps;
    if (p_end > npix) p_end = npix;
    const int niter = p_end > p_begin ? (int)((p_end - p_begin) / (4 * U)) : 0;     // pps and npix are multiples of 4U

    f32x4 acc[XTiles][GTiles];
#pragma unroll
    for (int r = 0; r < XTiles; ++r)
#pragma unroll
        for (int q = 0; q < GTiles; ++q) acc[r][q] = zero4();
    float bsum[GTiles];
#pragma unroll
    for (int q = 0; q < GTiles; ++q) bsum[q] = 0.f;

    int pj, pi, pn;                       // (column, row, image) of the current group's first pixel
    {
        const unsigned q = niter ? (unsigned)p_begin : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj = (int)(q - r * (unsigned)g.Wl);
        pn = (int)(r / (unsigned)g.Hl);
        pi = (int)(r - (unsigned)pn * (unsigned)g.Hl);
        // (the divisions run on the vector ALU: pin the uniform results to scalar registers)
        pj = __builtin_amdgcn_readfirstlane(pj); pi = __builtin_amdgcn_readfirstlane(pi); pn = __builtin_amdgcn_readfirstlane(pn);
    }
    const int dh = g.tdh[t], dw = g.tdw[t];
    const int xch = g.s_choff + cib * XB + (XV ? 4 * pl : pl);
    const int gch = g.d_choff + cob * GB + (GV ? 4 * pl : pl);
    // x: num_records = the tensor, so that the out-of-range offset LF_OOB reads as zero (the conv's padding); the launcher
    // keeps tensors below LF_OOB bytes
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)((long)g.N * g.Hs * g.Ws * g.s_pix << (S16 ? 1 : 2))),
                                 rg = make_rsrc(a.g, (unsigned)((long)g.N * g.Hd * g.Wd * g.d_pix << (S16 ? 1 : 2)));
    f32x4 psc, psh;
    float psc1[XTiles], psh1[XTiles];
    if (pro == LF_PRO_BNRELU) {
        if constexpr (XV) { psc = ldg4(a.pro_sc + cib * XB + 4 * pl); psh = ldg4(a.pro_sh + cib * XB + 4 * pl); }
        else {
#pragma unroll
            for (int r = 0; r < XTiles; ++r) { psc1[r] = a.pro_sc[cib * XB + r * 16 + pl]; psh1[r] = a.pro_sh[cib * XB + r * 16 + pl]; }
        }
    }

    // per-lane constants (element offsets inside a group) and the group strides
    const unsigned gvoff = (unsigned)(kq * g.dsw * g.d_pix + gch), gstep = (unsigned)(4 * g.dsw * g.d_pix);
    const unsigned xvoff = (unsigned)(kq * g.ssw * g.s_pix + xch), xstep = (unsigned)(4 * g.ssw * g.s_pix);
    constexpr unsigned ESH = S16 ? 1 : 2;          // element -> byte shift
    // One iteration = U k-steps = one group (the launcher guarantees Wl % (4U) == 0 and pps % (4U) == 0).  INTERIOR groups
    // (every tap position inside the image -- decided on scalars) load from a scalar base + per-lane constant; the others
    // compute per-lane offsets, and a tap position outside the image gets the out-of-range offset LF_OOB, which the buffer load
    // returns as 0.0f: no masks on the values.  Only with the BN+ReLU prologue (transform(0) != 0) the offsets are clamped
    // and the mask bits applied after the transform.
    // Loads are double-buffered in two named register sets: the next group's 2U loads are in flight during 16*U MFMAs.
    static_assert(!BFM || (S16 && XV && GV && U == 4), "BFM needs bf16 tensors, vector mode, U = 4");
    struct WStep {
        uint2 xr[BFM ? U : 1], gr[BFM ? U : 1];             // BFM: raw bf16 quads
        f32x4 x4[XV ? U : 1], g4[GV ? U : 1];
        float xs[XV ? 1 : U][XTiles], gs[GV ? 1 : U][GTiles];
        unsigned xmask;                                     // used with the prologue only
    };
    constexpr unsigned OOBE = LF_OOB >> ESH;                // in elements
    auto ldx = [&](WStep& S, int u, unsigned voff, unsigned soff) __attribute__((always_inline)) {      // element offsets
        if constexpr (BFM) S.xr[u] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(voff << 1), (int)(soff << 1), 0));
        else if constexpr (XV) {
            if constexpr (S16) S.x4[u] = epi_ld<true>(rx, voff, soff);
            else S.x4[u] = ldb4(rx, voff << 2, soff << 2);
        } else {
#pragma unroll
            for (int r = 0; r < XTiles; ++r) S.xs[u][r] = wg_ld1<S16>(rx, voff + r * 16, soff);
        }
    };
    auto wload = [&](WStep& S) __attribute__((always_inline)) {
        const unsigned gso = (unsigned)(((pn * g.Hd + pi * g.dsh + g.dah) * g.Wd + pj * g.dsw + g.daw) * g.d_pix);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned so = gso + u * gstep;
            if constexpr (BFM) S.gr[u] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rg, (int)(gvoff << 1), (int)(so << 1), 0));
            else if constexpr (GV) {
                if constexpr (S16) S.g4[u] = epi_ld<true>(rg, gvoff, so);
                else S.g4[u] = ldb4(rg, gvoff << 2, so << 2);
            } else {
#pragma unroll
                for (int q = 0; q < GTiles; ++q) S.gs[u][q] = wg_ld1<S16>(rg, gvoff + q * 16, so);
            }
        }
        const int sy = pi * g.ssh + dh;
        const bool yin = sy >= 0 && sy < g.Hs;
        const int sx0 = pj * g.ssw + dw, sxl = sx0 + (4 * U - 1) * g.ssw;
        if (yin && sx0 >= 0 && sxl < g.Ws) {
            const unsigned xso = (unsigned)(((pn * g.Hs + sy) * g.Ws + sx0) * g.s_pix);
#pragma unroll
            for (int u = 0; u < U; ++u) ldx(S, u, xvoff, xso + u * xstep);
            S.xmask = 0xfu;
        } else {
            const int syc = min(max(sy, 0), g.Hs - 1);
            const unsigned xso = (unsigned)((pn * g.Hs + syc) * g.Ws * g.s_pix);
            unsigned xm = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sx = sx0 + (kq + 4 * u) * g.ssw;
                const bool in = yin && sx >= 0 && sx < g.Ws;
                xm |= (in ? 1u : 0u) << u;
                const unsigned oc = (unsigned)(min(max(sx, 0), g.Ws - 1) * g.s_pix + xch);
                ldx(S, u, (in || pro == LF_PRO_BNRELU) ? oc : OOBE, xso);
            }
            S.xmask = xm;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        pj += 4 * U;
        if (pj >= g.Wl) { pj -= g.Wl; if (++pi >= g.Hl) { pi = 0; ++pn; } }
    };
    const bool need_bias = write_bias && a.bias_partial && t == 0 && cib == 0;      // workgroup-uniform
    // BIAS / PRO are compile-time in the loop body: left as scalar conditions the compiler if-converts them into v_cndmask
    // selects over BOTH sides -- the VALU work the scalar loop state was meant to remove.
    auto compute = [&](const WStep& S, auto BIAS_c, auto PRO_c) __attribute__((always_inline)) {
        constexpr bool BIAS = decltype(BIAS_c)::value, PRO = decltype(PRO_c)::value;
        if constexpr (BFM) {
            uint2 xq[U], gq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool xin = !PRO || ((S.xmask >> u) & 1u);
                uint2 xx = S.xr[u], gg = S.gr[u];
                if constexpr (PRO) {          // relu(bn(x)) in fp32 on the widened quad, rounded back
                    f32x4 t4;
                    t4.x = __uint_as_float(xx.x << 16); t4.y = __uint_as_float(xx.x & 0xffff0000u);
                    t4.z = __uint_as_float(xx.y << 16); t4.w = __uint_as_float(xx.y & 0xffff0000u);
                    t4 = max0(t4 * psc + psh);
                    lf_bf16x4 b;
                    b[0] = (lf_bf16)t4.x; b[1] = (lf_bf16)t4.y; b[2] = (lf_bf16)t4.z; b[3] = (lf_bf16)t4.w;
                    xx = __builtin_bit_cast(uint2, b);
                }
                xq[u].x = xin ? xx.x : 0u; xq[u].y = xin ? xx.y : 0u;
                gq[u] = gg;
            }
            // tile e of the x side = channels {4*row + e}: element e of every pixel's quad -> k = 4*kq + u
            s16x4 xa[4], gb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                uint2 ax, bx;
                if (e < 2) {
                    ax.x = __builtin_amdgcn_perm(xq[1].x, xq[0].x, sel); ax.y = __builtin_amdgcn_perm(xq[3].x, xq[2].x, sel);
                    bx.x = __builtin_amdgcn_perm(gq[1].x, gq[0].x, sel); bx.y = __builtin_amdgcn_perm(gq[3].x, gq[2].x, sel);
                } else {
                    ax.x = __builtin_amdgcn_perm(xq[1].y, xq[0].y, sel); ax.y = __builtin_amdgcn_perm(xq[3].y, xq[2].y, sel);
                    bx.x = __builtin_amdgcn_perm(gq[1].y, gq[0].y, sel); bx.y = __builtin_amdgcn_perm(gq[3].y, gq[2].y, sel);
                }
                xa[e] = __builtin_bit_cast(s16x4, ax);
                gb[e] = __builtin_bit_cast(s16x4, bx);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xa[r], gb[q], acc[r][q], 0, 0, 0);
            if constexpr (BIAS) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bsum[0] += __uint_as_float(gq[u].x << 16); bsum[1] += __uint_as_float(gq[u].x & 0xffff0000u);
                    bsum[2] += __uint_as_float(gq[u].y << 16); bsum[3] += __uint_as_float(gq[u].y & 0xffff0000u);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float xv[XTiles], gv[GTiles];
            if constexpr (GV) { gv[0] = S.g4[u].x; gv[1] = S.g4[u].y; gv[2] = S.g4[u].z; gv[3] = S.g4[u].w; }
            else {
#pragma unroll
                for (int q = 0; q < GTiles; ++q) gv[q] = S.gs[u][q];
            }
            if constexpr (XV) {
                f32x4 t4 = S.x4[u];
                if constexpr (PRO) t4 = max0(t4 * psc + psh);
                xv[0] = t4.x; xv[1] = t4.y; xv[2] = t4.z; xv[3] = t4.w;
            } else {
#pragma unroll
                for (int r = 0; r < XTiles; ++r) {
                    float t1 = S.xs[u][r];
                    if constexpr (PRO) t1 = fmaxf(t1 * psc1[r] + psh1[r], 0.f);
                    xv[r] = t1;
                }
            }
            if constexpr (PRO) {
                const bool xin = (S.xmask >> u) & 1u;
#pragma unroll
                for (int r = 0; r < XTiles; ++r) xv[r] = xin ? xv[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < XTiles; ++r)
#pragma unroll
                for (int q = 0; q < GTiles; ++q)
                    acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[r], gv[q], acc[r][q], 0, 0, 0);
            if constexpr (BIAS) {
#pragma unroll
                for (int q = 0; q < GTiles; ++q) bsum[q] += gv[q];
            }
        }
    };

    // (A third operand set -- loads two groups ahead -- spills: the three inlined loop bodies push the accumulators to scratch,
    // 185 us instead of 62.)
    WStep A, B;
    auto run = [&](auto BIAS_c, auto PRO_c) __attribute__((always_inline)) {
        auto step = [&](const WStep& S) __attribute__((always_inline)) { compute(S, BIAS_c, PRO_c); };
        wload(A);
        advance();
        if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tstamp[1] = __builtin_amdgcn_s_memrealtime(); }
        // Structured loop: pairs of groups (A then B), an odd last group peeled -- one loop body whose accumulators stay in place
        // (the first form, a for(;;) with two exits and conditional prefetches, made hipcc rename the 64 accumulator registers
        // from MFMA to MFMA and copy them at every join: 256 registers, 10 of them spilled in the prologue variant).
        // The prefetch behind the last group reads the next wave's first group -- or, past the tensors, zeros from the bounded
        // buffer resources -- and is never used.  Priority falls with progress (quarters): the waves of a SIMD finish together.
        const int npairs = niter >> 1, qp = npairs >> 2;
#pragma nounroll
        for (int ph = 0; ph < 4; ++ph) {
            switch (ph) {
                case 0: __builtin_amdgcn_s_setprio(3); break;
                case 1: __builtin_amdgcn_s_setprio(2); break;
                case 2: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
            const int n = ph < 3 ? qp : npairs - 3 * qp;
            for (int i = 0; i < n; ++i) {
                wload(B); advance();
                step(A);
                wload(A); advance();
                step(B);
            }
        }
        if (niter & 1) step(A);
    };
    if (niter > 0) {
        if constexpr (PROT >= 0) {
            if (need_bias) run(std::true_type{}, std::integral_constant<bool, PROT == 1>{});
            else run(std::false_type{}, std::integral_constant<bool, PROT == 1>{});
        } else {
            const bool prologue = pro == LF_PRO_BNRELU;
            if (need_bias) { if (prologue) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
            else { if (prologue) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }
        }
    }
    if constexpr (DBG) { asm volatile("" ::"v"(acc[0][0][0])); tstamp[2] = __builtin_amdgcn_s_memrealtime(); }
    // ---- reduce the 4 waves of the workgroup through LDS, wave 0 writes one partial row
    __builtin_amdgcn_s_setprio(3);
    __shared__ float red[WG_WAVES - 1][XTiles * GTiles * 4][64];
    __shared__ float bred[WG_WAVES][GTiles][64];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < XTiles; ++r)
#pragma unroll
            for (int q = 0; q < GTiles; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave - 1][(r * GTiles + q) * 4 + e][lane] = acc[r][q][e];
    }
#pragma unroll
    for (int q = 0; q < GTiles; ++q) bred[wave][q][lane] = bsum[q];
    __syncthreads();
    if (wave == 0) {
        // this workgroup's slab of the partial tensor: < 2^32 bytes (Cs * Cd <= 128 x 128), buffer-addressed
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.partial + ((long)bxs * g.ntaps + t) * g.Cs * g.Cd, 0xffffffffu);
#pragma unroll
        for (int r = 0; r < XTiles; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * kq + e;                                  // row of tile (r,q)
                const int ci = cib * XB + (XV ? 4 * i + r : r * 16 + i);
                float v[GTiles];
#pragma unroll
                for (int q = 0; q < GTiles; ++q) {
                    v[q] = acc[r][q][e];
#pragma unroll
                    for (int w = 0; w < WG_WAVES - 1; ++w) v[q] += red[w][(r * GTiles + q) * 4 + e][lane];
                }
                if constexpr (GV) {          // columns 4*pl .. 4*pl+3 of row ci: one 16-byte store
                    f32x4 v4 = {v[0], v[1], v[2], v[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v4), ro, (ci * g.Cd + cob * GB + 4 * pl) * 4, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < GTiles; ++q)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q]), ro, (ci * g.Cd + cob * GB + q * 16 + pl) * 4, 0, 0);
                }
            }
        if (write_bias && a.bias_partial && t == 0 && cib == 0) {
            // column sums of G over this workgroup's pixels: lanes with equal pl (4 k-slots) x 4 waves
#pragma unroll
            for (int q = 0; q < GTiles; ++q) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WG_WAVES; ++w) v += bred[w][q][lane];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (kq == 0) {
                    const int co = cob * GB + (GV ? 4 * pl + q : q * 16 + pl);
                    a.bias_partial[(long)bxs * g.Cd + co] = v;
                }
            }
        }
    }
    if constexpr (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.dbg) {
            unsigned long long* d = a.dbg + ((unsigned long long)blockIdx.x * WG_WAVES + wave) * 8;
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3];
            d[4] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
        }
    }
}

// 16 x 16 channel weight gradient with ALL taps in one wave (the 128x256 stage): one pass over G and X
// instead of one per tap -- these launches are HBM-bound, the per-tap split tripled their traffic.
template <int NTAPS, bool S16>
__global__ __launch_bounds__(256) void tapwgrad16_kernel(const LfTapGeom g, const LfWgradArgs a, const int pro,
                                                        const long pps, const int write_bias) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, kq = lane >> 4;
    const long npix = (long)g.N * g.Hl * g.Wl;
    const long sub = (long)blockIdx.x * WG_WAVES + wave;
    const long p_begin = sub * pps;
    long p_end = p_begin + pps;
    if (p_end > npix) p_end = npix;
    f32x4 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc[t] = zero4();
    float bsum = 0.f;
    float psc = 1.f, psh = 0.f;
    if (pro == LF_PRO_BNRELU) { psc = a.pro_sc[pl]; psh = a.pro_sh[pl]; }
    int tdh[NTAPS], tdw[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) { tdh[t] = g.tdh[t]; tdw[t] = g.tdw[t]; }
    long p = p_begin + kq;
    int pj, pi, pn;
    {
        const unsigned q = p < npix ? (unsigned)p : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj = (int)(q - r * (unsigned)g.Wl);
        pn = (int)(r / (unsigned)g.Hl);
        pi = (int)(r - (unsigned)pn * (unsigned)g.Hl);
    }
    const unsigned gstep = (unsigned)(4 * g.dsw * g.d_pix);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, 0xffffffffu), rg = make_rsrc(a.g, 0xffffffffu);
    while (p - kq < p_end) {                        // wave-uniform; 16 pixels of one row per iteration
        const unsigned gofs = (unsigned)(((pn * g.Hd + pi * g.dsh + g.dah) * g.Wd + pj * g.dsw + g.daw) * g.d_pix + g.d_choff + pl);
        float gv[U], xv[NTAPS][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool v = (p + 4 * u) < p_end;
            const float t0 = wg_ld1<S16>(rg, v ? gofs + u * gstep : gofs);
            gv[u] = v ? t0 : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int sy = pi * g.ssh + tdh[t];
            const bool yin = sy >= 0 && sy < g.Hs;
            const unsigned xrow = (unsigned)((pn * g.Hs + min(max(sy, 0), g.Hs - 1)) * g.Ws * g.s_pix + g.s_choff + pl);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sx = (pj + 4 * u) * g.ssw + tdw[t];
                const bool in = yin && sx >= 0 && sx < g.Ws && (p + 4 * u) < p_end;
                float x0 = wg_ld1<S16>(rx, xrow + (unsigned)(min(max(sx, 0), g.Ws - 1) * g.s_pix));
                if (pro == LF_PRO_BNRELU) x0 = fmaxf(x0 * psc + psh, 0.f);
                xv[t][u] = in ? x0 : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t][u], gv[u], acc[t], 0, 0, 0);
            bsum += gv[u];
        }
        p += 4 * U;
        pj += 4 * U;
        if (pj >= g.Wl) { pj -= g.Wl; if (++pi >= g.Hl) { pi = 0; ++pn; } }
    }
    __shared__ float red[WG_WAVES - 1][NTAPS * 4][64];
    __shared__ float bred[WG_WAVES][64];
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][t * 4 + e][lane] = acc[t][e];
    }
    bred[wave][lane] = bsum;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            float* out = a.partial + ((long)blockIdx.x * NTAPS + t) * g.Cs * g.Cd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[t][e];
#pragma unroll
                for (int w = 0; w < WG_WAVES - 1; ++w) v += red[w][t * 4 + e][lane];
                out[(long)(4 * kq + e) * g.Cd + pl] = v;          // row = x-channel 4*kq+e, col = g-channel pl
            }
        }
        if (write_bias && a.bias_partial) {
            float v = bred[0][lane] + bred[1][lane] + bred[2][lane] + bred[3][lane];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) a.bias_partial[(long)blockIdx.x * g.Cd + pl] = v;
        }
    }
}

// 16 x 16 channel weight gradient on bf16 tensors (no operand prologue): the kernel above fetches ONE bf16 per lane and
// instruction (2-byte buffer loads, 16 of them per 16 pixels: 135 us per launch at 160 x 320 x 64 images, where the 210 MB
// it reads take 40).  Here every wave owns a private LDS ring (no barrier in the loop):
//  * LDS-DMA copies 32 pixels x 32 bytes of G and of X at each of the three tap positions, lane-linear (1 KB = whole lines of the
//    NHWC tensors; a padding tap position carries the out-of-range offset and lands as zeros), three stages in flight;
//  * the image [pixel][16 channels] is exactly the [k][n] block ds_read_b64_tr_b16 transposes: lane (channel l & 15, k-group
//    l >> 4) receives pixels 4 (l >> 4) .. +3 of its channel -- the A / B operand of v_mfma_f32_16x16x16_bf16 with K = 16 pixels:
//    per 16 pixels one transposing read per operand, one MFMA per tap.
// Split over pixel ranges and reduced through LDS at the end like tapwgrad16_kernel (same partial layout).
// PRO (round 5): the BN + ReLU operand prologue of the block's third convolution on the transposed operand (a lane holds four pixels
// of ONE channel: scale / shift are two registers; padding re-zeroed after the transform) -- those two launches per step fell back
// to the 2-byte-load kernel before (137 us at 160 x 320 x 64 images).
constexpr int W16_STAGE = 4 * 1024, W16_STAGES = 4;
template <bool PRO>
__global__ __launch_bounds__(256) void tapwgrad16_tr_kernel(const LfTapGeom g, const LfWgradArgs a, const long pps, const int write_bias) {
    constexpr int NTAPS = 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pl = lane & 15, kq = lane >> 4;
    const long npix = (long)g.N * g.Hl * g.Wl;
    const long sub = (long)blockIdx.x * WG_WAVES + wave;
    const long p_begin = sub * pps;
    long p_end = p_begin + pps;
    if (p_end > npix) p_end = npix;
    const int niter = p_end > p_begin ? (int)((p_end - p_begin + 31) / 32) : 0;
    f32x4 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc[t] = zero4();
    float bsum = 0.f;
    const i32x4s rx = make_rsrc_words(a.x, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 2, (long)LF_OOB)),
                 rg = make_rsrc_words(a.g, (unsigned)min((long)g.N * g.Hd * g.Wd * g.d_pix * 2, (long)LF_OOB));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lf_tap_lds + (unsigned)(wave * W16_STAGES * W16_STAGE);
    const unsigned char* const ring = lf_tap_lds + wave * W16_STAGES * W16_STAGE;
    const int tdh0 = g.tdh[0], tdh1 = g.tdh[1], tdh2 = g.tdh[2], tdw0 = g.tdw[0], tdw1 = g.tdw[1], tdw2 = g.tdw[2];
    // load cursor: (image, row, column) of the next 16-pixel group (wave-uniform; Wl % 16 == 0: a group lies in one row)
    long p_ld = p_begin;
    int pj, pi, pn;
    {
        const unsigned q = niter ? (unsigned)p_begin : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj = __builtin_amdgcn_readfirstlane((int)(q - r * (unsigned)g.Wl));
        pn = __builtin_amdgcn_readfirstlane((int)(r / (unsigned)g.Hl));
        pi = __builtin_amdgcn_readfirstlane((int)r) - pn * g.Hl;
    }
    const int px = (lane >> 1) & 15, half = lane & 1;         // lanes 0-31: first group of the stage, 32-63: second
    int st_ld = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        // the two groups of this stage
        int gn[2], gi[2], gj[2];
        bool gok[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            gok[q] = p_ld < p_end;
            gn[q] = pn; gi[q] = pi; gj[q] = pj;
            p_ld += 16;
            pj += 16;
            if (pj >= g.Wl) { pj = 0; if (++pi >= g.Hl) { pi = 0; ++pn; } }
        }
        const bool hi = lane >= 32;
        const int n = hi ? gn[1] : gn[0], i = hi ? gi[1] : gi[0], j = (hi ? gj[1] : gj[0]) + px;
        const bool ok = hi ? gok[1] : gok[0];
        const unsigned dst = lds0 + (unsigned)(st_ld * W16_STAGE);
        const unsigned go = (unsigned)((((n * g.Hd + i * g.dsh + g.dah) * g.Wd + j * g.dsw + g.daw) * g.d_pix + g.d_choff + half * 8) * 2);
        lds_dma16(rg, dst, ok ? go : LF_OOB, 0u);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int dh = t == 0 ? tdh0 : t == 1 ? tdh1 : tdh2, dw = t == 0 ? tdw0 : t == 1 ? tdw1 : tdw2;
            const int sy = i * g.ssh + dh, sx = j * g.ssw + dw;
            const bool in = ok && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
            const unsigned xo = (unsigned)((((n * g.Hs + sy) * g.Ws + sx) * g.s_pix + g.s_choff + half * 8) * 2);
            lds_dma16(rx, dst + (unsigned)((1 + t) * 1024), in ? xo : LF_OOB, 0u);
        }
        st_ld = st_ld == W16_STAGES - 1 ? 0 : st_ld + 1;
    };
#pragma unroll
    for (int q = 0; q < W16_STAGES - 1; ++q) issue();
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
    float psc = 1.f, psh = 0.f;
    if constexpr (PRO) {
        psc = a.pro_sc[pl]; psh = a.pro_sh[pl];
        asm volatile("" ::"v"(psc), "v"(psh) : "memory");      // loaded before the loop's hand-counted vmcnt waits (see lf_wgrad_ro.hip)
    }
    // compute cursor (PRO: the padding tests of the group being multiplied): row and first column of its 16-pixel halves
    int cj = 0, ci = 0;
    {
        const unsigned q = niter ? (unsigned)p_begin : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        cj = __builtin_amdgcn_readfirstlane((int)(q - r * (unsigned)g.Wl));
        ci = __builtin_amdgcn_readfirstlane((int)(r % (unsigned)g.Hl));
    }
    int st_c = 0;
    for (int it = 0; it < niter; ++it) {
        // my DMA of this stage has landed (two younger stages of 4 instructions may be in flight); my reads of the stage that is
        // restaged next have retired.  The ring is private to the wave: no barrier.
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        issue();
        const unsigned char* st = ring + st_c * W16_STAGE;
        st_c = st_c == W16_STAGES - 1 ? 0 : st_c + 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const s16x4 gv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(st + q * 512 + lane * 8));
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                s16x4 xv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(st + (1 + t) * 1024 + q * 512 + lane * 8));
                if constexpr (PRO) {
                    const int dh = t == 0 ? tdh0 : t == 1 ? tdh1 : tdh2, dw = t == 0 ? tdw0 : t == 1 ? tdw1 : tdw2;
                    const int sy = ci * g.ssh + dh, c0 = cj * g.ssw + dw;            // the half's first pixel at this tap
                    if ((unsigned)sy >= (unsigned)g.Hs) { const s16x4 z = {0, 0, 0, 0}; xv = z; }
                    else {
                        const uint2 u = __builtin_bit_cast(uint2, xv);
                        const f32x2 sc2 = {psc, psc}, sh2 = {psh, psh};
                        f32x2 lo = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)}, hi = {__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                        lo = __builtin_elementwise_fma(lo, sc2, sh2);
                        hi = __builtin_elementwise_fma(hi, sc2, sh2);
                        const s16x2 z2 = {0, 0};
                        s16x2 a2 = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(lo, bf16x2)), z2);
                        s16x2 c2 = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(hi, bf16x2)), z2);
                        if (c0 < 0 || c0 + 15 * g.ssw >= g.Ws) {                    // edge half (wave-uniform test)
                            const int col = c0 + 4 * kq * g.ssw;
                            a2.x = (unsigned)(col) < (unsigned)g.Ws ? a2.x : (short)0; a2.y = (unsigned)(col + g.ssw) < (unsigned)g.Ws ? a2.y : (short)0;
                            c2.x = (unsigned)(col + 2 * g.ssw) < (unsigned)g.Ws ? c2.x : (short)0; c2.y = (unsigned)(col + 3 * g.ssw) < (unsigned)g.Ws ? c2.y : (short)0;
                        }
                        const s16x4 o = {a2.x, a2.y, c2.x, c2.y};
                        xv = o;
                    }
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xv, gv, acc[t], 0, 0, 0);
            }
            if constexpr (PRO) { cj += 16; if (cj >= g.Wl) { cj = 0; if (++ci >= g.Hl) ci = 0; } }
            bsum += __uint_as_float((unsigned)(unsigned short)gv.x << 16) + __uint_as_float((unsigned)(unsigned short)gv.y << 16) +
                    __uint_as_float((unsigned)(unsigned short)gv.z << 16) + __uint_as_float((unsigned)(unsigned short)gv.w << 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing (dead) stages: nothing may land in LDS after this
    __shared__ float red[WG_WAVES - 1][NTAPS * 4][64];
    __shared__ float bred[WG_WAVES][64];
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][t * 4 + e][lane] = acc[t][e];
    }
    bred[wave][lane] = bsum;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            float* out = a.partial + ((long)blockIdx.x * NTAPS + t) * g.Cs * g.Cd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[t][e];
#pragma unroll
                for (int w = 0; w < WG_WAVES - 1; ++w) v += red[w][t * 4 + e][lane];
                out[(long)(4 * kq + e) * g.Cd + pl] = v;          // row = x-channel 4*kq+e, col = g-channel pl
            }
        }
        if (write_bias && a.bias_partial) {
            float v = bred[0][lane] + bred[1][lane] + bred[2][lane] + bred[3][lane];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) a.bias_partial[(long)blockIdx.x * g.Cd + pl] = v;
        }
    }
}

// 16 x 16 channel weight gradient on fp32 tensors through LDS (round 5): tapwgrad16_kernel fetches ONE float per lane and instruction
// -- 16 vector-memory instructions of 256 bytes per 16 pixels; the launch sits on the address units at 43.6 us where its 134 MB take
// 24.  Same ring as the bf16 kernel above, per wave and without a barrier: LDS-DMA copies 16 pixels x 64 bytes of G and of X at
// each of the three tap positions as ONE 1 KB instruction each (lane-linear: pixel l >> 2, 16-byte chunk l & 3; a padding position
// carries the out-of-range offset and lands as zeros), and the MFMA operands -- lane (channel l & 15, pixel slot l >> 4) -- are
// plain ds_read_b32 of 256 contiguous bytes (conflict-free, 2 LDS cycles).  K = 4 pixels per v_mfma_f32_16x16x4_f32, 16 pixels
// per stage.  PRO: BN + ReLU on x in registers (one channel per lane), padding re-zeroed after the transform.
constexpr int W16F_STAGE = 4 * 1024, W16F_STAGES = 4;
template <bool PRO>
__global__ __launch_bounds__(256) void tapwgrad16_f32_kernel(const LfTapGeom g, const LfWgradArgs a, const long pps, const int write_bias) {
    constexpr int NTAPS = 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pl = lane & 15, kq = lane >> 4;
    const long npix = (long)g.N * g.Hl * g.Wl;
    const long sub = (long)blockIdx.x * WG_WAVES + wave;
    const long p_begin = sub * pps;
    long p_end = p_begin + pps;
    if (p_end > npix) p_end = npix;
    const int niter = p_end > p_begin ? (int)((p_end - p_begin + 15) / 16) : 0;
    f32x4 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc[t] = zero4();
    float bsum = 0.f;
    const i32x4s rx = make_rsrc_words(a.x, (unsigned)min((long)g.N * g.Hs * g.Ws * g.s_pix * 4, (long)LF_OOB)),
                 rg = make_rsrc_words(a.g, (unsigned)min((long)g.N * g.Hd * g.Wd * g.d_pix * 4, (long)LF_OOB));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lf_tap_lds + (unsigned)(wave * W16F_STAGES * W16F_STAGE);
    typedef __attribute__((address_space(3))) float* lds_f32_p;
    lds_f32_p const frag = (lds_f32_p)((__attribute__((address_space(3))) unsigned char*)lf_tap_lds + wave * W16F_STAGES * W16F_STAGE + kq * 64 + pl * 4);
    const int tdh0 = g.tdh[0], tdh1 = g.tdh[1], tdh2 = g.tdh[2], tdw0 = g.tdw[0], tdw1 = g.tdw[1], tdw2 = g.tdw[2];
    // load cursor: (image, row, column) of the next 16-pixel group (wave-uniform; Wl % 16 == 0: a group lies in one row)
    long p_ld = p_begin;
    int pj, pi, pn;
    {
        const unsigned q = niter ? (unsigned)p_begin : 0u;
        const unsigned r = q / (unsigned)g.Wl;
        pj = __builtin_amdgcn_readfirstlane((int)(q - r * (unsigned)g.Wl));
        pn = __builtin_amdgcn_readfirstlane((int)(r / (unsigned)g.Hl));
        pi = __builtin_amdgcn_readfirstlane((int)r) - pn * g.Hl;
    }
    int cj = pj, ci = pi;                                     // compute cursor (PRO: the padding tests)
    const int px = lane >> 2, chunk = lane & 3;
    auto issue = [&](const int slot) __attribute__((always_inline)) {
        const bool ok = p_ld < p_end;
        const int j = pj + px;
        const unsigned dst = lds0 + (unsigned)(slot * W16F_STAGE);
        const unsigned go = (unsigned)((((pn * g.Hd + pi * g.dsh + g.dah) * g.Wd + j * g.dsw + g.daw) * g.d_pix + g.d_choff + chunk * 4) * 4);
        lds_dma16(rg, dst, ok ? go : LF_OOB, 0u);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int dh = t == 0 ? tdh0 : t == 1 ? tdh1 : tdh2, dw = t == 0 ? tdw0 : t == 1 ? tdw1 : tdw2;
            const int sy = pi * g.ssh + dh, sx = j * g.ssw + dw;
            const bool in = ok && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
            const unsigned xo = (unsigned)((((pn * g.Hs + sy) * g.Ws + sx) * g.s_pix + g.s_choff + chunk * 4) * 4);
            lds_dma16(rx, dst + (unsigned)((1 + t) * 1024), in ? xo : LF_OOB, 0u);
        }
        p_ld += 16;
        pj += 16;
        if (pj >= g.Wl) { pj = 0; if (++pi >= g.Hl) { pi = 0; ++pn; } }
    };
    float psc = 1.f, psh = 0.f;
    if constexpr (PRO) {
        psc = a.pro_sc[pl]; psh = a.pro_sh[pl];
        asm volatile("" ::"v"(psc), "v"(psh) : "memory");      // returned before the hand-counted vmcnt waits of the loop
    }
#pragma unroll
    for (int q = 0; q < W16F_STAGES - 1; ++q) issue(q);
    for (int it0 = 0; it0 < niter; it0 += W16F_STAGES) {
#pragma unroll
        for (int sl = 0; sl < W16F_STAGES; ++sl) {
            if (it0 + sl >= niter) break;
            // my DMA of this stage has landed (two younger stages of 4 instructions may be in flight); my reads of the stage that is
            // restaged next have retired.  The ring is private to the wave: no barrier.
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            issue((sl + W16F_STAGES - 1) % W16F_STAGES);
            float gv[4], xv[NTAPS][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gv[u] = frag[(sl * W16F_STAGE + u * 256) / 4];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) xv[t][u] = frag[(sl * W16F_STAGE + (1 + t) * 1024 + u * 256) / 4];
            }
            if constexpr (PRO) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int dh = t == 0 ? tdh0 : t == 1 ? tdh1 : tdh2, dw = t == 0 ? tdw0 : t == 1 ? tdw1 : tdw2;
                    const bool rowok = (unsigned)(ci * g.ssh + dh) < (unsigned)g.Hs;
                    const int c0 = cj * g.ssw + dw;
                    const bool edge = c0 < 0 || c0 + 15 * g.ssw >= g.Ws;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float v = fmaxf(xv[t][u] * psc + psh, 0.f);
                        if (edge) v = (unsigned)(c0 + (4 * u + kq) * g.ssw) < (unsigned)g.Ws ? v : 0.f;
                        xv[t][u] = rowok ? v : 0.f;
                    }
                }
                cj += 16;
                if (cj >= g.Wl) { cj = 0; if (++ci >= g.Hl) ci = 0; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t][u], gv[u], acc[t], 0, 0, 0);
                bsum += gv[u];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing (dead) stages: nothing may land in LDS after this
    __shared__ float red[WG_WAVES - 1][NTAPS * 4][64];
    __shared__ float bred[WG_WAVES][64];
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][t * 4 + e][lane] = acc[t][e];
    }
    bred[wave][lane] = bsum;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            float* out = a.partial + ((long)blockIdx.x * NTAPS + t) * g.Cs * g.Cd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[t][e];
#pragma unroll
                for (int w = 0; w < WG_WAVES - 1; ++w) v += red[w][t * 4 + e][lane];
                out[(long)(4 * kq + e) * g.Cd + pl] = v;          // row = x-channel 4*kq+e, col = g-channel pl
            }
        }
        if (write_bias && a.bias_partial) {
            float v = bred[0][lane] + bred[1][lane] + bred[2][lane] + bred[3][lane];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) a.bias_partial[(long)blockIdx.x * g.Cd + pl] = v;
        }
    }
}

struct WgradCfg { int xv, gv, xt, gt, gx, u; long pps; };

WgradCfg wgrad_cfg(const LfTapGeom& g) {
    WgradCfg c;
    c.xv = (g.Cs % 64 == 0);
    c.gv = (g.Cd % 64 == 0);
    c.xt = c.xv ? 4 : g.Cs / 16;
    c.gt = c.gv ? 4 : g.Cd / 16;
    const int xb = c.xt * 16, gb = c.gt * 16;
    const bool small16 = (g.Cs == 16 && g.Cd == 16 && g.ntaps == 3 && g.Wl % 16 == 0);   // all taps in one wave
    const int jobs = small16 ? 1 : g.ntaps * (g.Cs / xb) * (g.Cd / gb);
    const long npix = (long)g.N * g.Hl * g.Wl;
    int gx = (small16 ? 4096 : 2048) / (jobs * WG_WAVES);   // 2 waves/SIMD (254 VGPRs) over 256 CUs; 8 for the lean 16x16 kernel
    if (gx < 1) gx = 1;
    const long maxgx = (npix + 64 * WG_WAVES - 1) / (64 * WG_WAVES);   // at least 64 pixels per wave
    if (gx > maxgx) gx = (int)maxgx;
    if (gx < 1) gx = 1;
    if (!small16) {   // make gx * jobs a multiple of 8 so the kernel's per-XCD ordering applies
        int mlt = 8;
        for (int d = 2; d <= 8; d *= 2) if (jobs % d == 0) mlt = 8 / d;
        if (gx >= mlt) gx = gx / mlt * mlt;
    }
    c.u = (g.Wl % 16 == 0) ? 4 : 1;           // k-steps per loop iteration (16 pixels of one row; 8-pixel iterations at 3 waves/SIMD spill: 1.7x slower)
    const int gran = 4 * c.u;
    long pps = (npix + (long)gx * WG_WAVES - 1) / ((long)gx * WG_WAVES);
    pps = (pps + gran - 1) / gran * gran;
    c.gx = (int)((npix + pps * WG_WAVES - 1) / (pps * WG_WAVES));
    c.pps = pps;
    return c;
}


}  // namespace

int lf_tapwgrad_splits(const LfTapGeom& g) { return wgrad_cfg(g).gx; }
int lf_tapwgrad_splits_bound(const LfTapGeom& g, int s16) {
    const int a = lf_tapwgrad_splits(g), c = s16 ? lf_tapwgrad_ro_rows_bound(g) : 0;
    return a > c ? a : c;
}
int lf_tapwgrad_splits_for(const LfTapGeom& g, const LfWgradArgs& a, int pro) {
    if (lf_tapwgrad_ro_ok(g, a.s16) && !a.dbg) return lf_tapwgrad_ro_rows(g);
    (void)pro;
    return wgrad_cfg(g).gx;
}

int lf_tapwgrad_launch(const LfTapGeom& g, const LfWgradArgs& a, int pro, hipStream_t st) {
    LF_REQUIRE(g.Cs % 16 == 0 && g.Cd % 16 == 0, "tapwgrad: channels must be multiples of 16");
    LF_REQUIRE(g.Wl % 4 == 0, "tapwgrad: logical width %d must be a multiple of 4", g.Wl);
    LF_REQUIRE((long)g.N * g.Hs * g.Ws * g.s_pix < (long)(LF_OOB >> 2) && (long)g.N * g.Hd * g.Wd * g.d_pix < (long)(LF_OOB >> 2),
               "tapwgrad: tensor too large for 32-bit byte offsets");
    if (lf_tapwgrad_ro_ok(g, a.s16) && !a.dbg) return lf_tapwgrad_ro_launch(g, a, pro, st);     // bf16 tensors, 3 taps, 64 / 128 channels
    const int wb = a.bias_partial != nullptr;
    const WgradCfg c = wgrad_cfg(g);
    const int xb = c.xt * 16, gb = c.gt * 16;
    if (g.Cs == 16 && g.Cd == 16 && g.ntaps == 3 && g.Wl % 16 == 0) {
        // both LDS-ring forms need 16-byte-aligned pixels of both tensors inside 32-bit byte offsets, and the LDS attribute
        const bool ring_ok = g_bf16_lds >= 3 && g.s_pix % 8 == 0 && g.s_choff % 8 == 0 && g.d_pix % 8 == 0 && g.d_choff % 8 == 0 &&
                             (long)g.N * g.Hs * g.Ws * g.s_pix * (a.s16 ? 2 : 4) < (long)LF_OOB && (long)g.N * g.Hd * g.Wd * g.d_pix * (a.s16 ? 2 : 4) < (long)LF_OOB;
        const size_t ring_lds = (size_t)WG_WAVES * W16_STAGES * W16_STAGE;
        static_assert(W16F_STAGES * W16F_STAGE == W16_STAGES * W16_STAGE, "the two 16-channel rings share their LDS budget");
        const bool pr = pro == LF_PRO_BNRELU;
#define LF_W16(KERN) do { if (allow_big_lds(reinterpret_cast<const void*>(KERN), 128 * 1024)) { hipLaunchKernelGGL(KERN, dim3(c.gx), dim3(256), ring_lds, st, g, a, c.pps, wb); launched = true; } } while (0)
        bool launched = false;
        if (ring_ok && a.s16) { if (pr) LF_W16(tapwgrad16_tr_kernel<true>); else LF_W16(tapwgrad16_tr_kernel<false>); }
        else if (ring_ok) { if (pr) LF_W16(tapwgrad16_f32_kernel<true>); else LF_W16(tapwgrad16_f32_kernel<false>); }
#undef LF_W16
        if (!launched) {           // (attribute refused, unaligned channel layout): the one-element-per-lane kernel
            if (a.s16) hipLaunchKernelGGL((tapwgrad16_kernel<3, true>), dim3(c.gx), dim3(256), 0, st, g, a, pro, c.pps, wb);
            else hipLaunchKernelGGL((tapwgrad16_kernel<3, false>), dim3(c.gx), dim3(256), 0, st, g, a, pro, c.pps, wb);
        }
        LF_CHECK_LAUNCH("tapwgrad16");
        return 0;
    }
    dim3 grid(c.gx * g.ntaps * (g.Cs / xb) * (g.Cd / gb));
    if (a.dbg) {           // phase stamps (tools/kbench.py --phases --wgrad): the fp32 64-channel-block kernel without prologue only
        LF_REQUIRE(c.xv && c.gv && c.u == 4 && !a.s16 && pro == LF_PRO_NONE, "tapwgrad: phase stamps are compiled into the plain fp32 kernel only");
        hipLaunchKernelGGL((tapwgrad_kernel<true, true, 4, 4, 4, false, false, 0, true>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx);
        LF_CHECK_LAUNCH("tapwgrad (stamps)");
        return 0;
    }
#define LF_WG(XV, GV, XT, GT)                                                                                     \
    do {                                                                                                          \
        if (a.s16 && c.u == 4 && XV && GV) hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 4, true, (XV && GV)>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx); \
        else if (a.s16 && c.u == 4) hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 4, true>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx); \
        else if (a.s16) hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 1, true>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx);   \
        else if (c.u == 4 && XV && GV && pro == LF_PRO_BNRELU) hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 4, false, false, (XV && GV) ? 1 : -1>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx); \
        else if (c.u == 4 && XV && GV) hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 4, false, false, (XV && GV) ? 0 : -1>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx); \
        else if (c.u == 4) hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 4, false>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx); \
        else hipLaunchKernelGGL((tapwgrad_kernel<XV, GV, XT, GT, 1, false>), grid, dim3(256), 0, st, g, a, pro, c.pps, wb, c.gx);  \
    } while (0)
    if (c.xv && c.gv) LF_WG(true, true, 4, 4);
    else if (c.xv && c.gt == 1) LF_WG(true, false, 4, 1);
    else if (c.xv && c.gt == 3) LF_WG(true, false, 4, 3);
    else if (!c.xv && c.xt == 1 && c.gv) LF_WG(false, true, 1, 4);
    else if (!c.xv && c.xt == 1 && c.gt == 1) LF_WG(false, false, 1, 1);
    else if (!c.xv && c.xt == 1 && c.gt == 3) LF_WG(false, false, 1, 3);
    else if (!c.xv && c.xt == 3 && c.gt == 1) LF_WG(false, false, 3, 1);
    else if (!c.xv && c.xt == 3 && c.gv) LF_WG(false, true, 3, 4);
    else return lf_fail("tapwgrad: unsupported channel combination Cs=%d Cd=%d", g.Cs, g.Cd);
#undef LF_WG
    LF_CHECK_LAUNCH("tapwgrad");
    return 0;
}

// ---------------------------------------------------------------------------------------
// split-K reduction + scatter into the PyTorch parameter-gradient layout; bias rows; packing
// ---------------------------------------------------------------------------------------
namespace {

struct TapIdx { int v[LF_MAX_TAPS]; };

__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int ntaps,
                                                           int Cs, int Cd, float* __restrict__ grad, long sk, long sn,
                                                           TapIdx ti, int wblocks, const float* __restrict__ brows,
                                                           int nbrows, float* __restrict__ bgrad, int baccum) {
    constexpr int SG = 16;                                       // 64 outputs x 16 split groups per block
    __shared__ float sw[SG][64];
    const int og = threadIdx.x & 63, sg = threadIdx.x >> 6;
    if ((int)blockIdx.x >= wblocks) {   // trailing blocks: bias gradient = column sums of the bias partial rows
        const int c = (blockIdx.x - wblocks) * 64 + og;
        float s = 0.f;
        if (c < Cd)
            for (int r = sg; r < nbrows; r += SG) s += brows[(long)r * Cd + c];
        sw[sg][og] = s;
        __syncthreads();
        if (sg == 0 && c < Cd) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < SG; ++k) v += sw[k][og];
            bgrad[c] = baccum ? bgrad[c] + v : v;
        }
        return;
    }
    const long per = (long)ntaps * Cs * Cd;
    for (long base = (long)blockIdx.x * 64; base < per; base += (long)wblocks * 64) {
        const long i = base + og;
        float s0 = 0.f, s1 = 0.f;
        if (i < per) {
            int r = sg;
            for (; r + SG < splits; r += 2 * SG) { s0 += partial[(long)r * per + i]; s1 += partial[(long)(r + SG) * per + i]; }
            for (; r < splits; r += SG) s0 += partial[(long)r * per + i];
        }
        sw[sg][og] = s0 + s1;
        __syncthreads();
        if (sg == 0 && i < per) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < SG; ++k) v += sw[k][og];          // fixed order: deterministic
            const int n = (int)(i % Cd);
            const long r2 = i / Cd;
            const int k2 = (int)(r2 % Cs), t = (int)(r2 / Cs);
            grad[k2 * sk + n * sn + ti.v[t]] = v;
        }
        __syncthreads();
    }
}

struct ReduceBatch { LfReduceJob j[LF_REDUCE_BATCH]; int blk0[LF_REDUCE_BATCH + 1]; };

// One workgroup = 64 outputs x 4 split groups; a thread sums every 4th partial row in four independent chains, the
// groups are combined through LDS in a fixed order (deterministic; not the summation order of wgrad_reduce_kernel).
// The workgroups of a job come first for its ntaps*Cs*Cd outputs, then one per 64 bias columns.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch B) {
    __shared__ float sw[4][64];
    int jb = 0;
    while (jb + 1 < LF_REDUCE_BATCH && (int)blockIdx.x >= B.blk0[jb + 1]) ++jb;      // uniform scan, <= 32 entries
    const LfReduceJob& J = B.j[jb];
    const int blk = blockIdx.x - B.blk0[jb];
    const int og = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long per = (long)J.ntaps * J.Cs * J.Cd;
    const int wblocks = (int)((per + 63) / 64);
    const float* __restrict__ rows;
    long width, i;
    int nrows;
    if (blk < wblocks) { rows = J.partial; width = per; i = (long)blk * 64 + og; nrows = J.splits; }
    else { rows = J.bias_rows; width = J.Cd; i = (long)(blk - wblocks) * 64 + og; nrows = J.n_bias_rows; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < width) {
        int r = sg;
        for (; r + 12 < nrows; r += 16) {
            s0 += rows[(long)r * width + i]; s1 += rows[(long)(r + 4) * width + i];
            s2 += rows[(long)(r + 8) * width + i]; s3 += rows[(long)(r + 12) * width + i];
        }
        for (; r < nrows; r += 4) s0 += rows[(long)r * width + i];
    }
    sw[sg][og] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0 && i < width) {
        const float v = (sw[0][og] + sw[1][og]) + (sw[2][og] + sw[3][og]);
        if (blk < wblocks) {
            const int n = (int)(i % J.Cd);
            const long r2 = i / J.Cd;
            const int k2 = (int)(r2 % J.Cs), t = (int)(r2 / J.Cs);
            J.grad[k2 * J.sk + n * J.sn + J.tapidx[t]] = v;
        } else {
            J.bias_grad[i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ rows, int nrows, int C,
                                                         float* __restrict__ dst, int accumulate) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    __shared__ float sb[4][64];
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int r = rg;
        for (; r + 4 < nrows; r += 8) { s0 += rows[(long)r * C + c]; s1 += rows[(long)(r + 4) * C + c]; }
        for (; r < nrows; r += 4) s0 += rows[(long)r * C + c];
    }
    sb[rg][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < C) {
        const float v = (sb[0][threadIdx.x] + sb[1][threadIdx.x]) + (sb[2][threadIdx.x] + sb[3][threadIdx.x]);
        dst[c] = accumulate ? dst[c] + v : v;
    }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const LfPackEntry* __restrict__ entries,
                                                          const float* const* __restrict__ params,
                                                          float* __restrict__ arena) {
    const LfPackEntry e = entries[blockIdx.x];
    const float* w = params[e.param];
    float* dst = arena + e.dst_off;
    // (32-bit index arithmetic: an entry has at most 9 * 128 * 128 elements, and the 64-bit divisions of the first version expanded
    // to ~100 instructions each -- 34 us for 2 M floats)
    const unsigned total = (unsigned)(e.ntaps * e.Kc * e.Nc), Nc = (unsigned)e.Nc, kbn = (unsigned)(e.Kc >> 2);
    for (unsigned i = blockIdx.y * 256u + threadIdx.x; i < total; i += gridDim.y * 256u) {
        const unsigned k4 = i & 3u, r = i >> 2, r2 = r / Nc, n = r - r2 * Nc, t = r2 / kbn, kb = r2 - t * kbn;
        const unsigned k = kb * 4u + k4;
        dst[i] = w[(long)k * e.sk + (long)n * e.sn + e.tapidx[t]];
    }
}

// bf16 operand order of tapgemm_bf16_kernel: [tap][ceil(Kc/32)*4][Nc][8], zero beyond Kc
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const LfPackEntry* __restrict__ entries,
                                                               const float* const* __restrict__ params,
                                                               __bf16* __restrict__ arena) {
    const LfPackEntry e = entries[blockIdx.x];
    const float* w = params[e.param];
    __bf16* dst = arena + e.dst16_off;
    const int kb_per_tap = ((e.Kc + 31) >> 5) * 4;
    const unsigned total = (unsigned)(e.ntaps * kb_per_tap * e.Nc * 8), Nc = (unsigned)e.Nc, kbn = (unsigned)kb_per_tap;     // (32-bit: see pack_weights_kernel)
    for (unsigned i = blockIdx.y * 256u + threadIdx.x; i < total; i += gridDim.y * 256u) {
        const unsigned k8 = i & 7u, r = i >> 3, r2 = r / Nc, n = r - r2 * Nc, t = r2 / kbn, kb = r2 - t * kbn;
        const int k = (int)(kb * 8u + k8);
        dst[i] = (__bf16)(k < e.Kc ? w[(long)k * e.sk + (long)n * e.sn + e.tapidx[t]] : 0.f);
    }
}

// split operand order of tapgemm_split_kernel: [tap][Kc/8][Nc][3][8] bf16, pieces h, m, l of every weight (exact:
// h + m + l == w); entries whose Kc is not a multiple of 32 are never used by that kernel and skipped here
__global__ __launch_bounds__(256) void pack_weights_split_kernel(const LfPackEntry* __restrict__ entries,
                                                                const float* const* __restrict__ params,
                                                                __bf16* __restrict__ arena) {
    const LfPackEntry e = entries[blockIdx.x];
    if (e.Kc % 32 != 0) return;
    const float* w = params[e.param];
    __bf16* dst = arena + 3 * e.dst16_off;
    const int kb_per_tap = e.Kc >> 3;
    const long total = (long)e.ntaps * kb_per_tap * e.Nc * 8;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const int k8 = (int)(i & 7);
        long r = i >> 3;
        const int n = (int)(r % e.Nc);
        const long row = r;                       // (t * kb_per_tap + kb) * Nc + n
        r /= e.Nc;
        const int kb = (int)(r % kb_per_tap);
        const int t = (int)(r / kb_per_tap);
        const float v = w[(kb * 8 + k8) * e.sk + n * e.sn + e.tapidx[t]];
        const __bf16 vh = (__bf16)v;
        const float r1 = v - (float)vh;
        const __bf16 vm = (__bf16)r1;
        const float r2 = r1 - (float)vm;
        dst[row * 24 + k8] = vh; dst[row * 24 + 8 + k8] = vm; dst[row * 24 + 16 + k8] = (__bf16)r2;
    }
}

}  // namespace

int lf_wgrad_reduce_launch(const float* partial, int splits, int ntaps, int Cs, int Cd, float* grad, long sk, long sn,
                           const int* tapidx_host, const float* bias_rows, int n_bias_rows, float* bias_grad,
                           int bias_accumulate, hipStream_t st) {
    TapIdx ti;
    for (int i = 0; i < LF_MAX_TAPS; ++i) ti.v[i] = i < ntaps ? tapidx_host[i] : 0;
    const long per = (long)ntaps * Cs * Cd;
    int wblocks = lf_cdiv(per, 64);
    if (wblocks > 4096) wblocks = 4096;
    const int bblocks = (bias_rows && bias_grad) ? lf_cdiv(Cd, 64) : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks + bblocks), dim3(1024), 0, st, partial, splits, ntaps, Cs, Cd, grad,
                       sk, sn, ti, wblocks, bias_rows, n_bias_rows, bias_grad, bias_accumulate);
    LF_CHECK_LAUNCH("wgrad_reduce");
    return 0;
}

int lf_wgrad_reduce_batch_launch(const LfReduceJob* jobs_host, int njobs, hipStream_t st) {
    for (int j0 = 0; j0 < njobs; j0 += LF_REDUCE_BATCH) {
        const int n = njobs - j0 < LF_REDUCE_BATCH ? njobs - j0 : LF_REDUCE_BATCH;
        ReduceBatch B;
        memset(&B, 0, sizeof(B));
        int blocks = 0;
        for (int i = 0; i < LF_REDUCE_BATCH; ++i) {
            B.blk0[i] = blocks;
            if (i >= n) continue;
            const LfReduceJob& J = jobs_host[j0 + i];
            B.j[i] = J;
            blocks += lf_cdiv((long)J.ntaps * J.Cs * J.Cd, 64) + ((J.bias_rows && J.bias_grad) ? lf_cdiv(J.Cd, 64) : 0);
        }
        B.blk0[LF_REDUCE_BATCH] = blocks;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, B);
        LF_CHECK_LAUNCH("wgrad_reduce_batch");
    }
    return 0;
}

int lf_rows_reduce_launch(const float* rows, int nrows, int C, float* dst, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(lf_cdiv(C, 64)), dim3(256), 0, st, rows, nrows, C, dst, accumulate);
    LF_CHECK_LAUNCH("rows_reduce");
    return 0;
}

int lf_pack_weights_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, float* arena,
                           hipStream_t st) {
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nentries, 16), dim3(256), 0, st, entries_dev, params_dev, arena);
    LF_CHECK_LAUNCH("pack_weights");
    return 0;
}

int lf_pack_weights_bf16_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, void* arena16,
                                hipStream_t st) {
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(nentries, 16), dim3(256), 0, st, entries_dev, params_dev,
                       reinterpret_cast<__bf16*>(arena16));
    LF_CHECK_LAUNCH("pack_weights_bf16");
    return 0;
}

int lf_pack_weights_split_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, void* arena48,
                                 hipStream_t st) {
    hipLaunchKernelGGL(pack_weights_split_kernel, dim3(nentries, 16), dim3(256), 0, st, entries_dev, params_dev,
                       reinterpret_cast<__bf16*>(arena48));
    LF_CHECK_LAUNCH("pack_weights_split");
    return 0;
}

long lf_pack_bf16_elems(int Kc, int Nc, int ntaps) { return (long)ntaps * (((Kc + 31) >> 5) * 32) * Nc; }
