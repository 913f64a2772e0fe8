// LDS-tiled tap-GEMM for the 64- / 128-channel layers (94 % of the network's MACs), fp32 matrix cores.
//
// Same contract as tapgemm_kernel (lf_conv.h): dst[dpix(p)][co] = epi(bias + sum_t sum_ci pro(src[spix(p,t)][ci]) * W[t][ci][co]),
// NHWC fp32, any tap table / strides; restricted to Cs % 32 == 0 and Cd == 64 or 128.
//
// Why a second kernel: tapgemm_kernel streams both operands L2 -> VGPR in fragment shape (16 pixels x 64 B per wave
// instruction: half cache lines, one address per lane); the texture-address path and the VGPR return port then cost ~25 %
// of the matrix-issue slots wherever the data sits (tools/mfma_rate.hip), and every wave re-reads the weights.  Here a
// workgroup stages full 128-byte lines ONCE per (tap, 32-channel chunk) with LDS-DMA (global_load_lds_dwordx4: no VGPRs,
// no per-element address VALU) and its four waves read MFMA fragments with conflict-free ds_read_b128.
//
// Geometry.  Workgroup = 4 waves; output tile = TP pixels x TC (= Cd) channels, wave tile 64 channels x 64 pixels as
// 2 x 2 tiles of v_mfma_f32_32x32x2_f32 (A = weights, rows = output channels; B = pixels, columns): TC = 128 -> waves
// 2 (channels) x 2 (pixels), TP = 128; TC = 64 -> 1 x 4, TP = 256.  A K-chunk = one tap x 32 source channels:
//   X chunk  [TP rows][8 slots of 16 B]   (a row = one pixel's 32 channels = one 128-B line)
//   W chunk  [TC rows][8 slots of 16 B]   (a row = one output channel's 32 k values; weights pre-packed [tap][Cs/32][Cd][32])
// both stored with slot' = slot ^ ((row >> 1) & 7): the LDS image of an LDS-DMA is lane-linear, so the XOR goes on the
// per-lane SOURCE address; a fragment read (lane = row l&31, k-half l>>5, slots 2q + (l>>5)) then touches 16 different
// 16-byte bank groups per 16-lane service group.  Two stages (64 / 80 KB): the DMA of chunk i+1 flies during the 64 MFMAs
// (4096 matrix cycles per wave) of chunk i; one workgroup barrier per chunk.  Padding: the DMA reads a clamped address and
// the fragment is masked after the read (the BN+ReLU prologue needs that order anyway).
// Workgroups are persistent over tiles: the stores of tile k drain while tile k+1 computes, BatchNorm partial sums stay in
// registers across tiles and leave as ONE row per workgroup (<= 512 rows per launch instead of one per 256 pixels).
// D layout (32x32): lane l holds pixel l&31, channels 8g + 4(l>>5) + {0..3} in registers 4g..4g+3 -> float4 stores.
#include "lf_conv.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ f32x4 max0(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}
__device__ __forceinline__ f32x4 keep_pos(f32x4 v, f32x4 m) {
    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ f32x4 sel4(bool c, f32x4 v) {
    v.x = c ? v.x : 0.f; v.y = c ? v.y : 0.f; v.z = c ? v.z : 0.f; v.w = c ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int TC>
struct LdsCfg {
    static constexpr int WN = TC / 64;            // wave blocks over output channels
    static constexpr int WM = 4 / WN;             // wave blocks over pixels
    static constexpr int TP = WM * 64;            // pixels per tile
    static constexpr int XI = TP / 32;            // X DMA instructions per wave and chunk (8 rows each)
    static constexpr int WI = TC / 32;            // W DMA instructions per wave and chunk
    static constexpr int SUB = (TP + TC) * 32;    // floats per 32-channel sub-chunk: X rows, then W rows
    static constexpr int STAGE = 2 * SUB;         // a stage = one tap x 64 source channels
};

// tap offsets by a select chain: indexing the kernel-argument arrays with a run-time tap makes hipcc copy them to scratch
__device__ __forceinline__ void tap_of(const LfTapGeom& g, int t, int& dh, int& dw) {
    dh = g.tdh[0]; dw = g.tdw[0];
#pragma unroll
    for (int k = 1; k < LF_MAX_TAPS; ++k)
        if (t == k) { dh = g.tdh[k]; dw = g.tdw[k]; }
}

struct PixCoord { int n, y, x; };
__device__ __forceinline__ PixCoord decompose(unsigned p, unsigned npix, int Hl, int Wl) {
    const unsigned q = p < npix ? p : 0u;
    const unsigned r = q / (unsigned)Wl;
    PixCoord c;
    c.x = (int)(q - r * (unsigned)Wl);
    c.n = (int)(r / (unsigned)Hl);
    c.y = (int)(r - (unsigned)c.n * (unsigned)Hl);
    return c;
}
__device__ __forceinline__ void advance(PixCoord& c, int step, int Hl, int Wl) {      // step <= Wl
    c.x += step;
    if (c.x >= Wl) { c.x -= Wl; if (++c.y >= Hl) { c.y = 0; ++c.n; } }
}

// The LDS (2 stages = 128 / 160 KB: one workgroup per CU) is DYNAMIC so that hipcc budgets registers by the launch bounds
// (256 per wave) alone: told the real LDS size it plans for one wave per SIMD, parks the accumulators in AGPRs and copies all
// 64 of them to VGPRs and back around every chunk (measured: 79 instead of 54 us for the 128-channel launch).
extern __shared__ __attribute__((aligned(16))) float lf_lds_dyn[];   // the ONLY LDS object (stages; statistics reduction at the end)

template <int TC, int PROC, bool STATS>
__global__ __launch_bounds__(256, 2) void tapgemm_lds_kernel(const LfTapGeom g, const LfTapArgs a, const int epi, const int ntiles) {
    using C = LdsCfg<TC>;
    float* const lds = lf_lds_dyn;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wn = wave % C::WN, wm = wave / C::WN;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl);
    const int nch = g.Cs >> 6;             // 64-channel chunks per tap
    const int ntaps = g.ntaps;
    // tools/kbench.py --phases: per-wave wall-clock stamps (s_memrealtime, 100 MHz) and ablation switches in the high bits of
    // `epi` (65536 = no stores, 131072 = no MFMAs, 262144 = only the first chunk is loaded); 0 in the product path
    unsigned long long tstamp[4] = {0ull, 0ull, 0ull, 0ull};
    if (a.dbg) tstamp[0] = __builtin_amdgcn_s_memrealtime();
    const bool ab_nostore = (epi & 65536) != 0, ab_nomma = (epi & 131072) != 0, ab_noload = (epi & 262144) != 0;

    // ---- tile schedule: workgroup b runs on XCD b % 8 (observed); every XCD gets a contiguous range of tiles so that the
    // halo rows neighbouring tiles share meet in one L2
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    int tile0, tstride, tend;
    if ((G & 7) == 0 && (ntiles & 7) == 0) {
        const int per = ntiles >> 3, gp = G >> 3, x = b & 7;
        tile0 = x * per + (b >> 3); tstride = gp; tend = (x + 1) * per;
    } else {
        tile0 = b; tstride = G; tend = ntiles;
    }

    // ---- per-lane constants
    const int swd = lane >> 4;                                   // DMA rows: (row >> 1) & 7 = ((i & 1) << 2) | (lane >> 4)
    int xslot[2];                                                // source slot (x4 floats) of this lane for even / odd DMA instruction
    xslot[0] = ((lane & 7) ^ swd) * 4;
    xslot[1] = ((lane & 7) ^ (swd | 4)) * 4;
    unsigned woff[C::WI];                                        // weights: float offset of this lane's 16 bytes inside a chunk
#pragma unroll
    for (int i = 0; i < C::WI; ++i) woff[i] = (unsigned)(((wave * C::WI + i) * 8 + (lane >> 3)) * 32 + xslot[i & 1]);
    const int swl = (lane >> 1) & 7;                             // fragment rows: (row >> 1) & 7
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fo[q] = ((2 * q + lh) ^ swl) * 4;
    const int aA = C::TP * 32 + (wn * 64 + l31) * 32;            // + i2 * 1024 + fo[q]
    const int aB = (wm * 64 + l31) * 32;                         // + j2 * 1024 + fo[q]

    // ---- per-tile state
    PixCoord xr[C::XI];                 // pixels of this lane's DMA rows (tile being loaded)
    unsigned xoff[C::XI];               // their source offsets for the tap being loaded
    struct TileInfo { unsigned vbits; unsigned dbase[2]; int pn[2]; bool pv[2]; };
    TileInfo cur, nxt;
    cur.vbits = 0; cur.dbase[0] = cur.dbase[1] = 0; cur.pn[0] = cur.pn[1] = 0; cur.pv[0] = cur.pv[1] = false;
    nxt = cur;

    auto prepare = [&](int tile) {      // DMA rows + fragment / epilogue pixels of `tile`
        PixCoord c = decompose((unsigned)tile * C::TP + (unsigned)(wave * C::XI * 8 + (lane >> 3)), npix, g.Hl, g.Wl);
#pragma unroll
        for (int i = 0; i < C::XI; ++i) { xr[i] = c; advance(c, 8, g.Hl, g.Wl); }
        const unsigned p0 = (unsigned)tile * C::TP + (unsigned)(wm * 64 + l31);
        PixCoord f = decompose(p0, npix, g.Hl, g.Wl);
        unsigned vb = 0;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const bool pv = p0 + 32u * j2 < npix;
            nxt.pv[j2] = pv; nxt.pn[j2] = f.n;
            nxt.dbase[j2] = (unsigned)(((f.n * g.Hd + f.y * g.dsh + g.dah) * g.Wd + f.x * g.dsw + g.daw) * g.d_pix + g.d_choff);
            for (int t = 0; t < ntaps; ++t) {
                int dh, dw;
                tap_of(g, t, dh, dw);
                const int sy = f.y * g.ssh + dh, sx = f.x * g.ssw + dw;
                const bool in = pv && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
                vb |= (in ? 1u : 0u) << (2 * t + j2);
            }
            advance(f, 32, g.Hl, g.Wl);
        }
        nxt.vbits = vb;
    };
    auto tap_offsets = [&](int t) {
        int dh, dw;
        tap_of(g, t, dh, dw);
#pragma unroll
        for (int i = 0; i < C::XI; ++i) {
            const int sy = min(max(xr[i].y * g.ssh + dh, 0), g.Hs - 1), sx = min(max(xr[i].x * g.ssw + dw, 0), g.Ws - 1);
            const int nn = min(xr[i].n, g.N - 1);                    // rows past the last pixel re-read image N-1 (never stored)
            xoff[i] = (unsigned)(((nn * g.Hs + sy) * g.Ws + sx) * g.s_pix + g.s_choff + xslot[i & 1]);
        }
    };
    auto issue = [&](int t, int ch, int stage) {
        if (ab_noload) return;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            float* sb = lds + stage * C::STAGE + sub * C::SUB;
#pragma unroll
            for (int i = 0; i < C::XI; ++i) glds16(a.src + xoff[i] + ch * 64 + sub * 32, sb + (wave * C::XI + i) * 256);
            const float* wsrc = a.wp32 + (long)((t * nch + ch) * 2 + sub) * (TC * 32);
#pragma unroll
            for (int i = 0; i < C::WI; ++i) glds16(wsrc + woff[i], sb + C::TP * 32 + (wave * C::WI + i) * 256);
        }
    };

    f32x16 acc[2][2];
    f32x4 s1[STATS ? 2 : 1][STATS ? 4 : 1], s2[STATS ? 2 : 1][STATS ? 4 : 1];
#pragma unroll
    for (int i2 = 0; i2 < (STATS ? 2 : 1); ++i2)
#pragma unroll
        for (int gq = 0; gq < (STATS ? 4 : 1); ++gq) { s1[i2][gq] = zero4(); s2[i2][gq] = zero4(); }

    auto clear_acc = [&]() {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i2][j2][r] = 0.f;
    };

    auto compute = [&](int t, int ch, int stage) {
        if (ab_nomma) return;
        const float* sb = lds + stage * C::STAGE;
        const bool v0 = (cur.vbits >> (2 * t)) & 1u, v1 = (cur.vbits >> (2 * t + 1)) & 1u;
        // 8 groups of 8 channels (two 32-channel sub-chunks); the fragments of group k+1 are requested before the 16 MFMAs of
        // group k (two register sets)
        f32x4 A0[2], A1[2], B0[2], B1[2];
        auto frag = [&](int k, int s) {
            const float* sq = sb + (k >> 2) * C::SUB;
            const int q = k & 3;
            A0[s] = *reinterpret_cast<const f32x4*>(sq + aA + fo[q]);
            A1[s] = *reinterpret_cast<const f32x4*>(sq + aA + 1024 + fo[q]);
            B0[s] = *reinterpret_cast<const f32x4*>(sq + aB + fo[q]);
            B1[s] = *reinterpret_cast<const f32x4*>(sq + aB + 1024 + fo[q]);
        };
        frag(0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cb = k & 1;
            if (k < 7) frag(k + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of this group's MFMAs (hipcc otherwise sinks them to their use)
            f32x4 b0 = B0[cb], b1 = B1[cb];
            if constexpr (PROC == LF_PRO_BNRELU) {
                // per-channel scale / shift of channels ch*64 + 8k + 4*lh + {0..3}: uniform addresses -> scalar loads (lgkmcnt,
                // not vmcnt: an ordinary vector load here would drain the DMA queue), selected by the lane's k-half
                const float* psc = a.pro_sc + ch * 64 + k * 8;
                const float* psh = a.pro_sh + ch * 64 + k * 8;
                f32x4 sc, sh;
                sc.x = lh ? psc[4] : psc[0]; sc.y = lh ? psc[5] : psc[1]; sc.z = lh ? psc[6] : psc[2]; sc.w = lh ? psc[7] : psc[3];
                sh.x = lh ? psh[4] : psh[0]; sh.y = lh ? psh[5] : psh[1]; sh.z = lh ? psh[6] : psh[2]; sh.w = lh ? psh[7] : psh[3];
                b0 = max0(b0 * sc + sh); b1 = max0(b1 * sc + sh);
            }
            b0 = sel4(v0, b0); b1 = sel4(v1, b1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cb][s], b0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cb][s], b1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb][s], b0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb][s], b1[s], acc[1][1], 0, 0, 0);
            }
        }
    };

    auto epilogue = [&]() {            // tile `cur`: bias, residual / masks, ReLU, store, statistics
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = wn * 64 + i2 * 32 + gq * 8 + lh * 4;
                const f32x4 bs = a.bias ? ldg4(a.bias + co) : zero4();
                f32x4 msc, msh, asc, ash;
                if (epi & LF_EPI_MASKBN) { msc = ldg4(a.msc + co); msh = ldg4(a.msh + co); }
                if (epi & LF_EPI_STATS_XHAT) { asc = ldg4(a.asc + co); ash = ldg4(a.ash + co); }
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    f32x4 v;
                    v.x = acc[i2][j2][gq * 4 + 0]; v.y = acc[i2][j2][gq * 4 + 1]; v.z = acc[i2][j2][gq * 4 + 2]; v.w = acc[i2][j2][gq * 4 + 3];
                    v += bs;
                    const unsigned off = cur.dbase[j2] + (unsigned)co;
                    f32x4 lx;
                    if (epi & LF_EPI_ADD) v += ldg4(a.add_src + off);
                    if (epi & LF_EPI_MASK) v = keep_pos(v, ldg4(a.mask_src + off));
                    if (epi & (LF_EPI_MASKBN | LF_EPI_STATS_XHAT)) lx = ldg4(a.aux + off);
                    if (epi & LF_EPI_MASKBN) v = keep_pos(v, lx * msc + msh);
                    if (epi & LF_EPI_RELU) v = max0(v);
                    if (cur.pv[j2] && !ab_nostore) *reinterpret_cast<f32x4*>(a.dst + off) = v;
                    if constexpr (STATS) {
                        v = sel4(cur.pv[j2], v);
                        if (epi & LF_EPI_STATS_SQ) { s1[i2][gq] += v; s2[i2][gq] += v * v; }
                        if (epi & LF_EPI_STATS_XHAT) {
                            const f32x4 gm = a.dm ? v * ldg4(a.dm + (long)cur.pn[j2] * g.Cd + co) : v;
                            s1[i2][gq] += gm; s2[i2][gq] += gm * (lx * asc + ash);
                        }
                    }
                }
            }
        }
    };

    // ---- flat pipeline over (tile, tap, chunk) items
    int tile = tile0;
    if (tile < tend) {
        prepare(tile);
        cur = nxt;
        tap_offsets(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {     // the first chunk (always loaded, also under the no-load ablation)
            float* sb = lds + sub * C::SUB;
#pragma unroll
            for (int i = 0; i < C::XI; ++i) glds16(a.src + xoff[i] + sub * 32, sb + (wave * C::XI + i) * 256);
#pragma unroll
            for (int i = 0; i < C::WI; ++i) glds16(a.wp32 + (long)sub * (TC * 32) + woff[i], sb + C::TP * 32 + (wave * C::WI + i) * 256);
        }
        clear_acc();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (a.dbg) tstamp[1] = __builtin_amdgcn_s_memrealtime();
        int t = 0, ch = 0, stage = 0;
        bool pending_epi = false;        // the previous tile's accumulators still await their epilogue
        for (;;) {
            // next item
            int nt = t, nc = ch + 1, ntile = tile;
            if (nc == nch) { nc = 0; if (++nt == ntaps) { nt = 0; ntile = tile + tstride; } }
            const bool has_next = ntile < tend;
            if (pending_epi) {           // first item of a tile: finish the previous one before touching the accumulators
                epilogue();
                clear_acc();
                cur = nxt;
                pending_epi = false;
            }
            if (has_next) {
                if (nc == 0) {
                    if (nt == 0) prepare(ntile);
                    tap_offsets(nt);
                }
                issue(nt, nc, stage ^ 1);
            }
            compute(t, ch, stage);
            if (nt == 0 && nc == 0) {    // that was the tile's last chunk
                if (!has_next) break;
                pending_epi = true;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            t = nt; ch = nc; tile = ntile; stage ^= 1;
        }
        if (a.dbg) {
            asm volatile("" ::"v"(acc[0][0][0]));
            tstamp[2] = __builtin_amdgcn_s_memrealtime();
        }
        epilogue();
        if (a.dbg) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tstamp[3] = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                unsigned long long* d = a.dbg + ((unsigned long long)b * 4 + wave) * 8;
                d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3];
            }
        }
    }

    if constexpr (STATS) {
        // one partial row per workgroup: red[v][thread] through the (now idle) stage memory, summed in a fixed order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = lds;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[(i2 * 16 + gq * 4 + e) * 256 + threadIdx.x] = s1[i2][gq][e];
                    red[(32 + i2 * 16 + gq * 4 + e) * 256 + threadIdx.x] = s2[i2][gq][e];
                }
        __syncthreads();
        if ((int)threadIdx.x < 2 * TC) {
            const int k = threadIdx.x / TC, c = threadIdx.x % TC;
            const int cwn = c >> 6, i2 = (c >> 5) & 1, gq = (c >> 3) & 3, clh = (c >> 2) & 1, e = c & 3;
            const float* row = red + (k * 32 + i2 * 16 + gq * 4 + e) * 256;
            float sum = 0.f;
            for (int w = 0; w < C::WM; ++w) {
                const float* rw = row + (w * C::WN + cwn) * 64 + clh * 32;
                for (int j = 0; j < 32; ++j) sum += rw[(j + threadIdx.x) & 31];      // rotated start: conflict-free, fixed order per thread
            }
            a.stats[((long)b * 2 + k) * g.Cd + c] = sum;
        }
    }
}

// weights in the order the kernel's DMA wants: wp32[((t*(Kc/32) + k/32)*Nc + n)*32 + k%32] = w[k*sk + n*sn + tapidx[t]]
__global__ __launch_bounds__(256) void pack_weights_lds_kernel(const LfPackEntry* __restrict__ entries,
                                                              const float* const* __restrict__ params,
                                                              float* __restrict__ arena) {
    const LfPackEntry e = entries[blockIdx.x];
    if (e.Kc % 32 != 0 || (e.Nc != 64 && e.Nc != 128)) return;
    const float* w = params[e.param];
    float* dst = arena + e.dst_off;
    const long total = (long)e.ntaps * e.Kc * e.Nc;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const int k32 = (int)(i & 31);
        long r = i >> 5;
        const int n = (int)(r % e.Nc);
        r /= e.Nc;
        const int kb = (int)(r % (e.Kc >> 5));
        const int t = (int)(r / (e.Kc >> 5));
        dst[i] = w[(kb * 32 + k32) * e.sk + n * e.sn + e.tapidx[t]];
    }
}

__global__ __launch_bounds__(256) void pack_one_lds_kernel(const float* __restrict__ w, float* __restrict__ dst, int Kc, int Nc,
                                                          int ntaps, long sk, long sn, int flip) {
    const long total = (long)ntaps * Kc * Nc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k32 = (int)(i & 31);
        long r = i >> 5;
        const int n = (int)(r % Nc);
        r /= Nc;
        const int kb = (int)(r % (Kc >> 5));
        const int t = (int)(r / (Kc >> 5));
        dst[i] = w[(kb * 32 + k32) * sk + n * sn + (flip ? ntaps - 1 - t : t)];
    }
}

int lds_tiles(const LfTapGeom& g) {
    const long npix = (long)g.N * g.Hl * g.Wl;
    const int tp = g.Cd == 128 ? LdsCfg<128>::TP : LdsCfg<64>::TP;
    return lf_cdiv(npix, tp);
}

}  // namespace

bool lf_tapgemm_lds_ok(const LfTapGeom& g) {
    return g.Cs % 64 == 0 && (g.Cd == 64 || g.Cd == 128) && g.s_pix % 4 == 0 && g.s_choff % 4 == 0 && g.d_pix % 4 == 0 &&
           g.d_choff % 4 == 0 && g.Wl >= 32 && (long)g.N * g.Hd * g.Wd * g.d_pix < (1L << 31);
}

// workgroups (= statistics rows) of a launch: persistent, one per CU (its four waves have a SIMD each)
int lf_tapgemm_lds_grid(const LfTapGeom& g) {
    const int nt = lds_tiles(g);
    return nt < 256 ? nt : 256;
}

int g_lds_ablate = 0;
void lf_tapgemm_lds_set_ablate(int mask) { g_lds_ablate = mask; }

int lf_tapgemm_lds_launch(const LfTapGeom& g, const LfTapArgs& a, int pro, int epi, hipStream_t st) {
    LF_REQUIRE(lf_tapgemm_lds_ok(g) && a.wp32, "tapgemm_lds: unsupported launch");
    epi |= g_lds_ablate << 16;
    const int nt = lds_tiles(g);
    const dim3 grid(lf_tapgemm_lds_grid(g));
    const bool stats = (epi & (LF_EPI_STATS_SQ | LF_EPI_STATS_XHAT)) != 0;
#define LF_TL1(TCV, PR, STV)                                                                                                \
    do {                                                                                                                    \
        constexpr unsigned bytes = 2u * LdsCfg<TCV>::STAGE * sizeof(float);                                                 \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tapgemm_lds_kernel<TCV, PR, STV>), \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);         \
        LF_REQUIRE(attr == hipSuccess, "tapgemm_lds: cannot reserve %u bytes of LDS: %s", bytes, hipGetErrorString(attr));  \
        hipLaunchKernelGGL((tapgemm_lds_kernel<TCV, PR, STV>), grid, dim3(256), bytes, st, g, a, epi, nt);                  \
    } while (0)
#define LF_TL(TCV)                                                                                                          \
    do {                                                                                                                    \
        if (pro == LF_PRO_BNRELU && stats) LF_TL1(TCV, 1, true);                                                            \
        else if (pro == LF_PRO_BNRELU) LF_TL1(TCV, 1, false);                                                               \
        else if (stats) LF_TL1(TCV, 0, true);                                                                               \
        else LF_TL1(TCV, 0, false);                                                                                         \
    } while (0)
    if (g.Cd == 128) LF_TL(128);
    else LF_TL(64);
#undef LF_TL
#undef LF_TL1
    LF_CHECK_LAUNCH("tapgemm_lds");
    return 0;
}

int lf_pack_weights_lds_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, float* arena32,
                               hipStream_t st) {
    hipLaunchKernelGGL(pack_weights_lds_kernel, dim3(nentries, 16), dim3(256), 0, st, entries_dev, params_dev, arena32);
    LF_CHECK_LAUNCH("pack_weights_lds");
    return 0;
}

int lf_pack_one_lds_launch(const float* w, float* dst, int Kc, int Nc, int ntaps, long sk, long sn, int flip, hipStream_t st) {
    hipLaunchKernelGGL(pack_one_lds_kernel, dim3(64), dim3(256), 0, st, w, dst, Kc, Nc, ntaps, sk, sn, flip);
    LF_CHECK_LAUNCH("pack_one_lds");
    return 0;
}
