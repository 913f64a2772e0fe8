// EXPERIMENT, NOT BUILT: the LDS-tiled, phase-locked 8-wave form of the tap-GEMM measured in round 2 (profiles/
// r2_kbench_lds_vs_streaming.txt, r2_lds_kernel_phases.txt).  It ties the shipped streaming kernel at 128 channels (60.1 vs
// 60.7 us warm) and loses at 64 (73.7 vs 66.7 us), so it was taken out of the library; commit 8c5b696 has it wired in
// (LfTapArgs::wp32 / zeros, lf_pack_weights_lds_launch, tools/kbench.py --variants 0).  Kept as the starting point for a
// 3-buffer / staggered-epilogue follow-up.
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
    static constexpr int WN = TC / 64;            // wave blocks over output channels (within a group)
    static constexpr int WM = 4 / WN;             // wave blocks over pixels (within a group)
    static constexpr int TPG = WM * 64;           // pixels per group tile; a workgroup's super-tile = 2 * TPG pixels
    static constexpr int XI = TPG / 32;           // X DMA instructions per wave and chunk (8 rows each)
    static constexpr int WI = TC / 32;            // W DMA instructions per group-A wave and chunk
    static constexpr int XS = TPG * 32;           // floats of one X chunk (one group, 32 channels)
    static constexpr int WS = TC * 32;            // floats of one W chunk
    static constexpr int OFF_W = 4 * XS;          // [X_A 0][X_A 1][X_B 0][X_B 1][W 0][W 1][W 2]
    static constexpr int TOTAL = 4 * XS + 3 * WS; // 112 KB (TC = 128) / 152 KB (TC = 64)
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

// The LDS is DYNAMIC so that hipcc budgets registers by the launch bounds (256 per wave) alone: told the real size (one
// workgroup per CU) it plans for the whole register file, parks the accumulators in AGPRs and copies all 64 of them to
// VGPRs and back around every chunk (measured on an earlier form of this kernel: 79 instead of 54 us per launch).
extern __shared__ __attribute__((aligned(16))) float lf_lds_dyn[];   // the ONLY LDS object

// Workgroup = 512 threads = two 4-wave groups A (waves 0-3) and B (waves 4-7); waves w and w+4 share SIMD w.  Each group
// owns one TPG-pixel tile of the workgroup's super-tile and walks the same (tap, 32-channel chunk) items, but B runs ONE
// PHASE behind A, phase-locked by the workgroup barrier: while one group streams the 64 MFMAs of an item (its "compute"
// phase, 4096 matrix cycles), the other is in its "service" phase -- epilogue of a finished tile, LDS-DMA requests for the
// item after next, fragment reads for its next item -- so every SIMD's matrix pipe always has exactly one wave issuing
// back-to-back MFMAs and nobody's stalls land in front of the pipe.  (Measured before this form: two independent 4-wave
// workgroups per CU, or one with a lone wave per SIMD, keep the pipes only 65-80 % busy: every wait of a wave idles its SIMD
// unless the partner happens to be ready, and the older workgroup wins every arbitration.)
// Buffers: X chunks are private to a group and double-buffered (a group refills the buffer it has just finished reading,
// during its service phase); the W chunk is shared (A loads it, both read it one phase apart) and needs three buffers.
// Every DMA is waited for (vmcnt(0)) by its issuer at the end of the compute phase that follows -- a full phase of lead --
// and published by that phase's barrier.
// EPI: the epilogue flags as a compile-time constant for the combinations the network uses (straight-line epilogue: all of a
// tile's operand loads in flight together), or -1 = read them from `epi_rt` (hipcc then branches around every load and
// waits for each one separately).
template <int TC, int PROC, int EPI>
__global__ __launch_bounds__(512, 2) void tapgemm_lds_kernel(const LfTapGeom g, const LfTapArgs a, const int epi_rt, const int nsuper) {
    using C = LdsCfg<TC>;
    const int epi = EPI >= 0 ? EPI : (epi_rt & 63);
    constexpr bool STATS = EPI < 0 || (EPI & (LF_EPI_STATS_SQ | LF_EPI_STATS_XHAT)) != 0;
    float* const lds = lf_lds_dyn;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, gw = wave & 3;                    // group (0 = A, 1 = B), wave within the group
    const int l31 = lane & 31, lh = lane >> 5;
    const int wn = gw % C::WN, wm = gw / C::WN;
    const unsigned npix = (unsigned)(g.N * g.Hl * g.Wl);
    const int nch = g.Cs >> 5;                                   // 32-channel chunks per tap
    const int ntaps = g.ntaps;
    // tools/kbench.py --phases: per-wave wall-clock stamps (s_memrealtime, 100 MHz) and ablation switches in the high bits of
    // `epi` (65536 = no stores, 131072 = no MFMAs, 262144 = no DMA after the first items); 0 in the product path
    unsigned long long tstamp[4] = {0ull, 0ull, 0ull, 0ull};
    if (a.dbg) tstamp[0] = __builtin_amdgcn_s_memrealtime();
    const bool ab_nostore = (epi_rt & 65536) != 0, ab_nomma = (epi_rt & 131072) != 0, ab_noload = (epi_rt & 262144) != 0;

    // ---- super-tile schedule: workgroup b runs on XCD b % 8 (observed); every XCD gets a contiguous range so that the halo
    // rows neighbouring tiles share meet in one L2
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    int st0, ststride, stend;
    if ((G & 7) == 0 && (nsuper & 7) == 0) {
        const int per = nsuper >> 3, gp = G >> 3, x = b & 7;
        st0 = x * per + (b >> 3); ststride = gp; stend = (x + 1) * per;
    } else {
        st0 = b; ststride = G; stend = nsuper;
    }
    const int ntile = (stend - st0 + ststride - 1) / ststride;  // tiles of this group (>= 1)
    const int nitems = ntile * ntaps * nch;

    // ---- per-lane constants
    const int swd = lane >> 4;                                   // DMA rows: (row >> 1) & 7 = ((i & 1) << 2) | (lane >> 4)
    int xslot[2];                                                // source slot (x4 floats) of this lane for even / odd DMA instruction
    xslot[0] = ((lane & 7) ^ swd) * 4;
    xslot[1] = ((lane & 7) ^ (swd | 4)) * 4;
    unsigned woff[C::WI];                                        // weights: float offset of this lane's 16 bytes inside a chunk
#pragma unroll
    for (int i = 0; i < C::WI; ++i) woff[i] = (unsigned)(((gw * C::WI + i) * 8 + (lane >> 3)) * 32 + xslot[i & 1]);
    const int swl = (lane >> 1) & 7;                             // fragment rows: (row >> 1) & 7
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fo[q] = ((2 * q + lh) ^ swl) * 4;
    const int aA = (wn * 64 + l31) * 32;                         // inside a W chunk: + i2 * 1024 + fo[q]
    const int aB = (wm * 64 + l31) * 32;                         // inside an X chunk: + j2 * 1024 + fo[q]
    float* const xbase = lds + grp * 2 * C::XS;                  // this group's two X buffers
    float* const wbase = lds + C::OFF_W;

    // ---- item cursors: (tile ordinal, tap, chunk); the load cursor runs one item ahead of the compute cursor
    struct Cursor { int tile, t, ch, idx; };
    auto step = [&](Cursor& c) {
        ++c.idx;
        if (++c.ch == nch) { c.ch = 0; if (++c.t == ntaps) { c.t = 0; ++c.tile; } }
    };

    // ---- per-tile state
    PixCoord xr[C::XI];                 // pixels of this lane's DMA rows (tile being loaded)
    unsigned xoff[C::XI];               // their source offsets for the tap being loaded
    unsigned xpad = 0;                  // bit i: row i of that tap is padding (PROC == 0: its DMA reads the zero page instead)
    struct TileInfo { unsigned vbits; unsigned dbase[2]; int pn[2]; bool pv[2]; };
    TileInfo cur, nxt;
    cur.vbits = 0; cur.dbase[0] = cur.dbase[1] = 0; cur.pn[0] = cur.pn[1] = 0; cur.pv[0] = cur.pv[1] = false;
    nxt = cur;

    auto prepare = [&](int tile_ord) {  // DMA rows + fragment / epilogue pixels of this group's tile number tile_ord
        const unsigned pbase = ((unsigned)(st0 + tile_ord * ststride) * 2u + (unsigned)grp) * C::TPG;
        PixCoord c = decompose(pbase + (unsigned)(gw * C::XI * 8 + (lane >> 3)), npix, g.Hl, g.Wl);
#pragma unroll
        for (int i = 0; i < C::XI; ++i) { xr[i] = c; advance(c, 8, g.Hl, g.Wl); }
        const unsigned p0 = pbase + (unsigned)(wm * 64 + l31);
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
        xpad = 0;
#pragma unroll
        for (int i = 0; i < C::XI; ++i) {
            const int syr = xr[i].y * g.ssh + dh, sxr = xr[i].x * g.ssw + dw;
            const int sy = min(max(syr, 0), g.Hs - 1), sx = min(max(sxr, 0), g.Ws - 1);
            const int nn = min(xr[i].n, g.N - 1);                    // rows past the last pixel re-read image N-1 (never stored)
            xoff[i] = (unsigned)(((nn * g.Hs + sy) * g.Ws + sx) * g.s_pix + g.s_choff + xslot[i & 1]);
            xpad |= ((sy != syr || sx != sxr) ? 1u : 0u) << i;
        }
    };
    // requests of item c: this group's X chunk (buffer idx & 1) and, from group A, the shared W chunk (buffer idx % 3)
    auto issue = [&](const Cursor& c) {
        if (c.ch == 0) {
            if (c.t == 0) prepare(c.tile);
            tap_offsets(c.t);
        }
        float* xb = xbase + (c.idx & 1) * C::XS;
#pragma unroll
        for (int i = 0; i < C::XI; ++i) {
            const float* src = a.src + xoff[i] + c.ch * 32;
            if constexpr (PROC == LF_PRO_NONE) src = ((xpad >> i) & 1u) ? a.zeros + xslot[i & 1] : src;     // padding reads zeros
            glds16(src, xb + (gw * C::XI + i) * 256);
        }
        if (grp == 0) {
            float* wb = wbase + (c.idx % 3) * C::WS;
            const float* wsrc = a.wp32 + (long)(c.t * nch + c.ch) * C::WS;
#pragma unroll
            for (int i = 0; i < C::WI; ++i) glds16(wsrc + woff[i], wb + (gw * C::WI + i) * 256);
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

    // fragments: two register sets; set 0 is filled in the service phase for group 0 of the coming item
    f32x4 A0[2], A1[2], B0[2], B1[2];
    auto frag = [&](int idx, int q, int s) {
        const float* wb = wbase + (idx % 3) * C::WS;
        const float* xb = xbase + (idx & 1) * C::XS;
        A0[s] = *reinterpret_cast<const f32x4*>(wb + aA + fo[q]);
        A1[s] = *reinterpret_cast<const f32x4*>(wb + aA + 1024 + fo[q]);
        B0[s] = *reinterpret_cast<const f32x4*>(xb + aB + fo[q]);
        B1[s] = *reinterpret_cast<const f32x4*>(xb + aB + 1024 + fo[q]);
    };
    // One item = 4 groups of 8 channels, 16 MFMAs each.  Order inside a group: 4 MFMAs, the 4 fragment requests of the NEXT group,
    // 12 MFMAs.  PROC == 0 needs no VALU at all (padding arrives as zeros); with the BN+ReLU prologue the next group's pixel
    // fragments are transformed between the last 8 MFMAs, four VALU per MFMA, so that no burst of vector work ever sits in front
    // of the matrix pipe (tools/mfma_mix.hip: 8 v_cndmask in a row ahead of every 16 MFMAs cost 9 %).
    auto transform = [&](const Cursor& c, int q, f32x4& b0, f32x4& b1, bool v0, bool v1) {
        // per-channel scale / shift of channels ch*32 + 8q + 4*lh + {0..3}: uniform addresses -> scalar loads (lgkmcnt, not
        // vmcnt: an ordinary vector load here would drain the DMA queue), selected by the lane's k-half
        const float* psc = a.pro_sc + c.ch * 32 + q * 8;
        const float* psh = a.pro_sh + c.ch * 32 + q * 8;
        f32x4 sc, sh;
        sc.x = lh ? psc[4] : psc[0]; sc.y = lh ? psc[5] : psc[1]; sc.z = lh ? psc[6] : psc[2]; sc.w = lh ? psc[7] : psc[3];
        sh.x = lh ? psh[4] : psh[0]; sh.y = lh ? psh[5] : psh[1]; sh.z = lh ? psh[6] : psh[2]; sh.w = lh ? psh[7] : psh[3];
        b0 = sel4(v0, max0(b0 * sc + sh)); b1 = sel4(v1, max0(b1 * sc + sh));
    };
    auto mma = [&](int cb, int s) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cb][s], B0[cb][s], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cb][s], B1[cb][s], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb][s], B0[cb][s], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cb][s], B1[cb][s], acc[1][1], 0, 0, 0);
    };
    auto compute = [&](const Cursor& c) {
        if (ab_nomma) return;
        const bool v0 = (cur.vbits >> (2 * c.t)) & 1u, v1 = (cur.vbits >> (2 * c.t + 1)) & 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = q & 1;
            mma(cb, 0);
            if (q < 3) frag(c.idx, q + 1, cb ^ 1);
            mma(cb, 1);
            if constexpr (PROC == LF_PRO_BNRELU) {
                if (q < 3) transform(c, q + 1, B0[cb ^ 1], B1[cb ^ 1], v0, v1);
            }
            mma(cb, 2);
            mma(cb, 3);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);            // 4 MFMA
            if (q < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); // 4 DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);            // 4 MFMA
            if constexpr (PROC == LF_PRO_BNRELU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);    // <= 5 VALU behind it
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto epilogue = [&]() {            // tile `cur`: bias, residual / masks, ReLU, store, statistics
        int opaque = 0;                 // keeps the per-channel vector loads INSIDE the epilogue: they are tile-invariant and
        asm volatile("" : "+s"(opaque));   // hipcc otherwise hoists all 8 x 5 of them out of the item loop (128 registers, spills)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = wn * 64 + i2 * 32 + gq * 8 + lh * 4 + opaque;
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
                // operand loads are batched over two channel groups (<= 12 float4 in flight): hoisting all of a tile's loads
                // to the top spills 100+ registers
                if (gq & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: items 0 of both groups
    Cursor cc = {0, 0, 0, 0}, lc = {0, 0, 0, 0};            // compute cursor, load cursor
    issue(lc);
    cur = nxt;
    step(lc);
    clear_acc();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.dbg) tstamp[1] = __builtin_amdgcn_s_memrealtime();
    if (grp == 1) __syncthreads();          // B runs one phase behind A (A is in its first service phase)

    // tools/kbench.py --phases: waves of the first 4 workgroups also log the time of every barrier (64 stamps per wave,
    // behind the 2048 x 8 summary words)
    unsigned long long* const trace = (a.dbg && b < 4 && lane == 0) ? a.dbg + 2048 * 8 + (b * 8 + wave) * 64 : nullptr;
    bool pending_epi = false;
    for (int j = 0; j < nitems; ++j) {
        // ---- service phase of item j (the partner group computes)
        if (pending_epi) {                  // the tile that ended with item j-1
            epilogue();
            clear_acc();
            cur = nxt;
            pending_epi = false;
        }
        if (lc.idx < nitems && !(ab_noload && lc.idx > 0)) issue(lc);     // item j+1 (its tile's coordinates go to `nxt`)
        if (lc.idx < nitems) step(lc);
        frag(cc.idx, 0, 0);
        if constexpr (PROC == LF_PRO_BNRELU)
            transform(cc, 0, B0[0], B1[0], (cur.vbits >> (2 * cc.t)) & 1u, (cur.vbits >> (2 * cc.t + 1)) & 1u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (trace && j < 31) trace[2 * j] = __builtin_amdgcn_s_memrealtime();
        // ---- compute phase of item j
        compute(cc);
        if (cc.ch == nch - 1 && cc.t == ntaps - 1) pending_epi = true;
        step(cc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my requests of the service phase above have landed ...
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                       // ... and are published
        __builtin_amdgcn_sched_barrier(0);
        if (trace && j < 31) trace[2 * j + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.dbg) {
        asm volatile("" ::"v"(acc[0][0][0]));
        tstamp[2] = __builtin_amdgcn_s_memrealtime();
    }
    epilogue();                              // the last tile
    if (grp == 0) __builtin_amdgcn_s_barrier();              // pairs B's extra first barrier (B's last compute phase)
    if (a.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* d = a.dbg + ((unsigned long long)b * 8 + wave) * 8;
            d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3];
        }
    }

    if constexpr (STATS) {
        // one partial row per workgroup: red[v][thread] through the (now idle) LDS, sums then sums of squares (64 KB each),
        // added in a fixed order
        const int tid = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            float* red = lds;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) red[(i2 * 16 + gq * 4 + e) * 512 + tid] = k == 0 ? s1[i2][gq][e] : s2[i2][gq][e];
            __syncthreads();
            if (tid < TC) {
                const int c = tid;
                const int cwn = c >> 6, i2 = (c >> 5) & 1, gq = (c >> 3) & 3, clh = (c >> 2) & 1, e = c & 3;
                const float* row = red + (i2 * 16 + gq * 4 + e) * 512;
                float sum = 0.f;
                for (int gr = 0; gr < 2; ++gr)
                    for (int w = 0; w < C::WM; ++w) {
                        const float* rw = row + (gr * 4 + w * C::WN + cwn) * 64 + clh * 32;
                        for (int jj = 0; jj < 32; ++jj) sum += rw[(jj + tid) & 31];      // rotated start: conflict-free, fixed order per thread
                    }
                a.stats[((long)b * 2 + k) * g.Cd + c] = sum;
            }
        }
    }
}

// weights in the order the kernel's DMA wants: wp32[((t*(Kc/32) + k/32)*Nc + n)*32 + k%32] = w[k*sk + n*sn + tapidx[t]]
__global__ __launch_bounds__(256) void pack_weights_lds_kernel(const LfPackEntry* __restrict__ entries,
                                                              const float* const* __restrict__ params,
                                                              float* __restrict__ arena, float* __restrict__ zeros) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) zeros[threadIdx.x] = 0.f;
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
                                                          int ntaps, long sk, long sn, int flip, float* __restrict__ zeros) {
    if (blockIdx.x == 0 && threadIdx.x < 64) zeros[threadIdx.x] = 0.f;
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

int lds_super_tiles(const LfTapGeom& g) {
    const long npix = (long)g.N * g.Hl * g.Wl;
    const int tpg = g.Cd == 128 ? LdsCfg<128>::TPG : LdsCfg<64>::TPG;
    return lf_cdiv(npix, 2 * tpg);
}

}  // namespace

bool lf_tapgemm_lds_ok(const LfTapGeom& g) {
    return g.Cs % 32 == 0 && (g.Cd == 64 || g.Cd == 128) && g.s_pix % 4 == 0 && g.s_choff % 4 == 0 && g.d_pix % 4 == 0 &&
           g.d_choff % 4 == 0 && g.Wl >= 32 && (long)g.N * g.Hd * g.Wd * g.d_pix < (1L << 31);
}

// workgroups (= statistics rows) of a launch: persistent, one 8-wave workgroup per CU
int lf_tapgemm_lds_grid(const LfTapGeom& g) {
    const int nt = lds_super_tiles(g);
    return nt < 256 ? nt : 256;
}

int g_lds_ablate = 0;
void lf_tapgemm_lds_set_ablate(int mask) { g_lds_ablate = mask; }

int lf_tapgemm_lds_launch(const LfTapGeom& g, const LfTapArgs& a, int pro, int epi, hipStream_t st) {
    LF_REQUIRE(lf_tapgemm_lds_ok(g) && a.wp32, "tapgemm_lds: unsupported launch");
    epi |= g_lds_ablate << 16;
    const int nt = lds_super_tiles(g);
    const dim3 grid(lf_tapgemm_lds_grid(g));
    LF_REQUIRE(a.zeros, "tapgemm_lds: zero page missing");
#define LF_TL1(TCV, PR, EP)                                                                                                 \
    do {                                                                                                                    \
        constexpr unsigned bytes = (unsigned)LdsCfg<TCV>::TOTAL * sizeof(float);                                            \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tapgemm_lds_kernel<TCV, PR, EP>), \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);         \
        LF_REQUIRE(attr == hipSuccess, "tapgemm_lds: cannot reserve %u bytes of LDS: %s", bytes, hipGetErrorString(attr));  \
        hipLaunchKernelGGL((tapgemm_lds_kernel<TCV, PR, EP>), grid, dim3(512), bytes, st, g, a, epi, nt);                   \
    } while (0)
    // the epilogue combinations of the network get a compiled-in epilogue; anything else takes the run-time form
#define LF_TL(TCV)                                                                                                          \
    do {                                                                                                                    \
        const int e = epi & 63;                                                                                             \
        if (pro == LF_PRO_BNRELU) { if (e == LF_EPI_RELU) LF_TL1(TCV, 1, LF_EPI_RELU); else LF_TL1(TCV, 1, -1); }           \
        else if (e == 0) LF_TL1(TCV, 0, 0);                                                                                 \
        else if (e == LF_EPI_RELU) LF_TL1(TCV, 0, LF_EPI_RELU);                                                             \
        else if (e == LF_EPI_STATS_SQ) LF_TL1(TCV, 0, LF_EPI_STATS_SQ);                                                     \
        else if (e == LF_EPI_MASK) LF_TL1(TCV, 0, LF_EPI_MASK);                                                             \
        else if (e == LF_EPI_ADD) LF_TL1(TCV, 0, LF_EPI_ADD);                                                               \
        else LF_TL1(TCV, 0, -1);      /* the BatchNorm-backward sums (STATS_XHAT): compiled-in flags spill there */ \
    } while (0)
    if (g.Cd == 128) LF_TL(128);
    else LF_TL(64);
#undef LF_TL
#undef LF_TL1
    LF_CHECK_LAUNCH("tapgemm_lds");
    return 0;
}

int lf_pack_weights_lds_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, float* arena32,
                               float* zeros, hipStream_t st) {
    hipLaunchKernelGGL(pack_weights_lds_kernel, dim3(nentries, 16), dim3(256), 0, st, entries_dev, params_dev, arena32, zeros);
    LF_CHECK_LAUNCH("pack_weights_lds");
    return 0;
}

int lf_pack_one_lds_launch(const float* w, float* dst, int Kc, int Nc, int ntaps, long sk, long sn, int flip, float* zeros,
                           hipStream_t st) {
    hipLaunchKernelGGL(pack_one_lds_kernel, dim3(64), dim3(256), 0, st, w, dst, Kc, Nc, ntaps, sk, sn, flip, zeros);
    LF_CHECK_LAUNCH("pack_one_lds");
    return 0;
}
