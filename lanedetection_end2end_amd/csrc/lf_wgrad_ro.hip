// Read-once weight gradient of the 3-tap C -> C convolutions on bf16 tensors (C = 64, 128; precision mode "bf16"):
//     dW[t][ci][co] = sum_p pro(X[p + tap t][ci]) * G[p][co]          (non_bottleneck_1d, BEV/Networks/ERFNet.py:29-37,44-60)
//
// Why: tapwgrad_kernel splits a launch into (tap, 64 x 64 channel block) jobs and every job re-streams its halves of X and G -- 3
// jobs at 64 channels, 12 at 128 -- so the launch sits on the L2 -> CU delivery rate times that redundancy (92 us for the 210 MB
// of an 80 x 160 x 64-image launch, 55 us for the 105 MB of the 128-channel one: DESIGN.md 4.3).  Here ONE workgroup owns all three
// taps of a 64-channel block of X against ALL channels of G for its pixels:
//   * workgroup = C / 16 waves; wave (r, gh) owns x-channels 16 r .. + 15 of the block, g-channels 64 gh .. + 63, 3 taps:
//     12 accumulator tiles of 16 x 16 (48 registers), K = 32 pixels per v_mfma_f32_16x16x32_bf16;
//   * operands travel by LDS-DMA in whole 128- / 256-byte lines (lf_ldsdma.h) into a ring of stages of TWO 16-pixel groups:
//     per group G [16 px][C] once and X [16 px][64] at each of the three tap positions (a padding position carries the out-of-range
//     offset and lands as zeros).  A wave has ONE role -- G, or X at one tap -- and issues four of the stage's 1 KB instructions;
//     a group whose tap positions lie inside the image (decided on scalars) is a scalar base + a per-lane constant;
//   * the [pixel][channel] image of a group IS the [k][n] block ds_read_b64_tr_b16 transposes into the MFMA operand (lane =
//     channel l & 15, pixels 4 (l >> 4) .. + 3): the halves of a K = 32 operand are the two groups' reads (k is permuted the same
//     way on both operands); 14 transposing reads feed a wave's 12 MFMAs; the 32-byte channel chunks of a pixel are XOR-swizzled
//     by the pixel index on the SOURCE side of the DMA so that the 8 pixels a half-wave reads fall into 8 different bank groups;
//   * the BN + ReLU operand prologue (the block's third convolution reads relu(bn1(t2)), never stored) is applied to the
//     transposed operand in registers: a lane holds ONE channel, so scale / shift are two registers; padding is re-zeroed
//     after the transform (wave-uniform row test, per-lane column test on edge groups only).
// The first form of this kernel (one group per stage, every wave staging a mix of tensors) was INSTRUCTION-bound: ~150 mostly scalar
// instructions per wave and 16 pixels, 47 us per launch with or without its DMA and MFMA instructions.
// At 128 channels the two 64-channel blocks of X are two workgroups on the same rows (G is read twice from L2, X once per
// tap): 20 KB through LDS per 16 pixels against 16 for a 16-wave workgroup owning everything, and half the partial rows per
// launch of a form that gives every workgroup the whole 3 x 128 x 128 result.
// No cross-wave reduction: every output has one owner; a workgroup writes its partial row [3][64 of C][C] (fp32), the existing
// split-K reduction sums the rows in a fixed order.  Bias gradient = column sums of G: one extra MFMA with an all-ones A
// operand per stage on the wave whose r names the g-tile.
#include <map>
#include <mutex>

#include "lf_conv.h"
#include "lf_ldsdma.h"
#include "lf_types.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
typedef __attribute__((address_space(3))) unsigned char* lds_u8_p;

extern __shared__ __attribute__((aligned(16))) unsigned char lf_ro_lds[];

template <int C> struct RoCfg {
    static constexpr int NW = C / 16;                  // waves per workgroup
    static constexpr int GBY = 16 * C * 2;             // bytes of G per 16-pixel group
    static constexpr int XBY = 16 * 64 * 2;            // bytes of one tap position of the 64-channel x-block
    static constexpr int GRP = GBY + 3 * XBY;          // a group's block: G, then X at the three tap positions (8 KB / 10 KB)
    static constexpr int STAGE = 2 * GRP;              // two groups = 32 pixels = one K = 32 MFMA step
    static constexpr int IPW = 4;                      // DMA instructions per staging wave and stage
    // Ring depth (even: the two operand register sets alternate with the unrolled ring): 64 channels 4 x 16 KB, two workgroups
    // per CU; 128 channels 6 x 20 KB, one 8-wave workgroup per CU -- ~100 KB in flight per CU either way.
    static constexpr int D = C == 64 ? 4 : 6;
    static constexpr int SPB = C == 64 ? 4 : 3;        // stages per fragment base pointer (a ds_read's immediate offset is 16 bits)
    static constexpr int NBASE = D / SPB;
    static constexpr size_t LDS = (size_t)D * STAGE;
    static_assert(D % 2 == 0 && D % SPB == 0 && LDS <= 160 * 1024, "ring layout");
    static_assert((SPB - 1) * STAGE + GRP + 2 * XBY < 65536, "fragment offsets must fit the immediate");
};

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// PRO: 0 none, 1 BN + ReLU on x with taps along H only (the network's case: no column tests), 2 ... with taps along W
template <int C, int PRO>
__global__ __launch_bounds__(C * 4, C == 64 ? 2 : 1) void tapwgrad_ro_kernel(const LfTapGeom g, const LfWgradArgs a,
                                                                            const int write_bias) {
    typedef RoCfg<C> K;
    constexpr int D = K::D, STAGE = K::STAGE, GRP = K::GRP, GBY = K::GBY, XBY = K::XBY, NXB = C / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = wave & 3, gh = wave >> 2;
    const int b = lane >> 4, j16 = lane & 15;
    // Work = image rows, dealt out so that what the vertical taps re-read is in the L2 when they ask for it.  Workgroup L runs on
    // XCD L % 8: XCD x owns a contiguous block of rows and its S workgroups take them ROUND ROBIN (rows w, w + S, ... of the block): at
    // any moment the XCD works on ~S consecutive rows, so the rows i +- d a workgroup's taps read are the rows its neighbours w +- d
    // stage as their own centre tap at the same time -- one L2 fill per line instead of three.  (Contiguous ranges per workgroup put
    // those re-reads one row apart in time in 64 workgroups per L2: the 3 x 1 convolution ran at 67 us where the 1 x 3 one, whose
    // taps hit the L1, took 52.)  The two x-blocks of a row at 128 channels are adjacent workgroups of the same XCD.
    unsigned ord = blockIdx.x;
    const bool by_xcd = (gridDim.x & 7u) == 0 && ((gridDim.x / (unsigned)NXB) & 7u) == 0;
    if (by_xcd) ord = (ord & 7u) * (gridDim.x >> 3) + (ord >> 3);
    const int cib = (int)(ord % (unsigned)NXB);
    const unsigned bxs = ord / (unsigned)NXB, nsplit = gridDim.x / (unsigned)NXB;
    const unsigned S = by_xcd ? nsplit >> 3 : nsplit;                     // workgroups (per x-block) sharing a row block
    const unsigned NR = (unsigned)(g.N * g.Hl);
    const unsigned RB = by_xcd ? (NR + 7u) >> 3 : NR;
    const unsigned xcd = bxs / S, wq = bxs - xcd * S;
    const unsigned row0 = xcd * RB + wq;
    unsigned row_end = (xcd + 1u) * RB;
    if (row_end > NR) row_end = NR;
    const unsigned nrows = row0 < row_end ? (row_end - row0 + S - 1u) / S : 0u;
    const int nstages = (int)((nrows * ((unsigned)g.Wl >> 4) + 1u) >> 1);  // workgroup-uniform

    f32x4 acc[3][4], bacc = zero4();
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = zero4();

    const int dh0 = g.tdh[0], dh1 = g.tdh[1], dh2 = g.tdh[2], dw0 = g.tdw[0], dw1 = g.tdw[1], dw2 = g.tdw[2];
    const unsigned lds0 = (unsigned)(size_t)(lds_u8_p)lf_ro_lds;

    // ---- this wave's staging role (wave-uniform): 64 channels: wave 0 = G of both groups, waves 1-3 = X at tap wave - 1;
    // 128 channels: wave 0 / 1 = G of group A / B, waves 2-4 = X at tap wave - 2, waves 5-7 stage nothing.  A staging wave issues
    // two UNITS of two instructions per stage; unit u serves group u (the 128-channel G waves: their group, pixel halves u).
    const int xw0 = C == 64 ? 1 : 2;
    const bool isg = wave < xw0, stager = wave < xw0 + 3;
    const int tap = wave - xw0;
    const int rdh = isg ? 0 : (tap == 0 ? dh0 : tap == 1 ? dh1 : dh2), rdw = isg ? 0 : (tap == 0 ? dw0 : tap == 1 ? dw1 : dw2);
    const int rpix = isg ? g.d_pix : g.s_pix, rpixb = rpix * 2;
    const i32x4s rs = isg ? make_rsrc_words(a.g, (unsigned)((long)g.N * g.Hd * g.Wd * g.d_pix * 2))
                          : make_rsrc_words(a.x, (unsigned)((long)g.N * g.Hs * g.Ws * g.s_pix * 2));
    const bool g128 = isg && C == 128;
    // instruction j of a unit: pixels PXJ * j + lpix (+ 8 u for the 128-channel G waves) of the group, 16-byte chunk `chunk` of them
    const int lpix = g128 ? lane >> 4 : lane >> 3;
    unsigned lane_v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = (g128 ? 4 : 8) * j + lpix;
        const int chunk = g128 ? (lane & 15) ^ (2 * (px & 7)) : (lane & 7) ^ (2 * ((px >> 1) & 3));
        const int choff = isg ? g.d_choff : g.s_choff + 64 * cib;
        lane_v[j] = (unsigned)((px * rpix + choff) * 2 + chunk * 16);
    }
    const unsigned rdst = isg ? (C == 128 ? (unsigned)(wave * GRP) : 0u) : (unsigned)(GBY + tap * XBY);   // inside a stage; + group / pixel-half offsets below
    const unsigned voob = LF_OOB;

    // cursor over this workgroup's groups: (image, row, first column), wave-uniform (Wl % 16 == 0: a group lies in one row)
    struct Cur { int n, i, j; unsigned row; };
    auto advance = [&](Cur& c) __attribute__((always_inline)) {
        c.j += 16;
        if (c.j >= g.Wl) {
            c.j = 0; c.row += S; c.i += (int)S;
            while (c.i >= g.Hl) { c.i -= g.Hl; ++c.n; }
        }
    };
    Cur ld;
    {
        const unsigned q = nrows ? row0 : 0u;
        ld.n = __builtin_amdgcn_readfirstlane((int)(q / (unsigned)g.Hl));
        ld.i = __builtin_amdgcn_readfirstlane((int)q) - ld.n * g.Hl;
        ld.j = 0; ld.row = row0;
    }
    Cur cc = ld;                                               // compute cursor (the prologue's padding tests)
    auto issue_unit = [&](const Cur& c, const unsigned dst, const int pxu) __attribute__((always_inline)) {
        // the unit's group at this wave's tap: row sy, columns c0 + pxu .. of the tensor
        const int sy = c.i + rdh, c0 = c.j + rdw;
        const bool yok = c.row < row_end && (unsigned)sy < (unsigned)g.Hl;
        const int base = (((c.n * g.Hl + sy) * g.Wl + c0) + pxu) * rpixb;
        if (yok && c0 >= 0 && c0 + 15 < g.Wl) {               // interior (scalar test): scalar base + per-lane constant
#pragma unroll
            for (int j = 0; j < 2; ++j) lds_dma16(rs, dst + (unsigned)j * 1024u, lane_v[j], (unsigned)base);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool in = yok && (unsigned)(c0 + pxu + (g128 ? 4 : 8) * j + lpix) < (unsigned)g.Wl;
                lds_dma16(rs, dst + (unsigned)j * 1024u, in ? (unsigned)base + lane_v[j] : voob, 0u);
            }
        }
    };
    auto issue = [&](const int slot) __attribute__((always_inline)) {
        const Cur A = ld;
        advance(ld);
        const Cur B = ld;
        advance(ld);
        if (stager) {
            const unsigned sb = lds0 + (unsigned)(slot * STAGE) + rdst;
            if (g128) {                                        // one group (A for wave 0, B for wave 1), its two pixel halves
                const Cur& G0 = wave == 0 ? A : B;
                issue_unit(G0, sb, 0);
                issue_unit(G0, sb + 2048u, 8);
            } else {
                issue_unit(A, sb, 0);
                issue_unit(B, sb + (unsigned)GRP, 0);
            }
        }
    };

    // ---- fragment addresses inside a group block (per lane): pixel p = 4 b + (j16 >> 2), 8 bytes at (j16 & 3) * 8 of the 32-byte chunk
    const int fp = 4 * b + (j16 >> 2), fsub = (j16 & 3) * 8;
    lds_u8_p const ldsp = (lds_u8_p)lf_ro_lds;
    lds_u8_p xfrag[K::NBASE], gfrag[K::NBASE][4];
#pragma unroll
    for (int h = 0; h < K::NBASE; ++h) {
        lds_u8_p const hb = ldsp + h * K::SPB * STAGE;
        xfrag[h] = hb + GBY + fp * 128 + ((r ^ ((fp >> 1) & 3)) << 5) + fsub;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            gfrag[h][q] = C == 64 ? hb + fp * 128 + ((q ^ ((fp >> 1) & 3)) << 5) + fsub
                                  : hb + fp * 256 + (((4 * gh + q) ^ (fp & 7)) << 5) + fsub;
    }

    float psc = 1.f, psh = 0.f;
    if constexpr (PRO) {
        psc = a.pro_sc[cib * 64 + r * 16 + j16]; psh = a.pro_sh[cib * 64 + r * 16 + j16];
        // the two loads must have returned before the first DMA: hipcc cannot see the asm's outstanding vector-memory operations and
        // would otherwise place `s_waitcnt vmcnt(0)` -- the whole ring drained -- in front of their first use INSIDE the loop
        asm volatile("" ::"v"(psc), "v"(psh) : "memory");
    }
    const bool need_bias = write_bias && a.bias_partial && cib == 0;      // workgroup-uniform
    const s16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0

    s16x4 xs[2][2][3], gs[2][2][4];                            // [register set][group][tap / g-tile]
    auto read_stage = [&](const int slot, const int set) __attribute__((always_inline)) {
        const int h = slot / K::SPB, so = (slot % K::SPB) * STAGE;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) gs[set][gi][q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(gfrag[h][q] + so + gi * GRP));
#pragma unroll
            for (int t = 0; t < 3; ++t) xs[set][gi][t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(xfrag[h] + so + gi * GRP + t * XBY));
        }
    };
    // relu(bn(x)) on a transposed operand half (four pixels of this lane's channel), rounded back to bf16; pixels outside the image
    // stay zero: (ci, cj) = row and first column of the half's group.  Packed arithmetic (every VALU instruction here is one more
    // issue slot of a loop that is instruction-bound): two v_pk_fma_f32, two v_cvt_pk_bf16_f32, and the ReLU as a signed 16-bit
    // maximum with 0 on the ROUNDED pairs (v_pk_max_i16) -- rounding is monotonic and keeps the sign, so relu(round(y)) ==
    // round(relu(y)) bit for bit, the value the forward pass's operand prologue formed.
    auto transform = [&](s16x4 v, const int ci, const int cj, const int dh, const int dw) __attribute__((always_inline)) -> s16x4 {
        if ((unsigned)(ci + dh) >= (unsigned)g.Hl) { const s16x4 z = {0, 0, 0, 0}; return z; }    // the whole tap row is padding
        const uint2 u = __builtin_bit_cast(uint2, v);
        const f32x2 sc2 = {psc, psc}, sh2 = {psh, psh};
        f32x2 lo = {bf_lo(u.x), bf_hi(u.x)}, hi = {bf_lo(u.y), bf_hi(u.y)};
        lo = __builtin_elementwise_fma(lo, sc2, sh2);
        hi = __builtin_elementwise_fma(hi, sc2, sh2);
        const s16x2 z2 = {0, 0};
        s16x2 a = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(lo, bf16x2)), z2);
        s16x2 c = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(hi, bf16x2)), z2);
        const int c0 = cj + dw;                                          // column of the group's first pixel at this tap
        if (PRO == 2 && (c0 < 0 || c0 + 15 >= g.Wl)) {                   // edge group (wave-uniform test)
            const int col = c0 + 4 * b;
            a.x = (unsigned)(col + 0) < (unsigned)g.Wl ? a.x : (short)0; a.y = (unsigned)(col + 1) < (unsigned)g.Wl ? a.y : (short)0;
            c.x = (unsigned)(col + 2) < (unsigned)g.Wl ? c.x : (short)0; c.y = (unsigned)(col + 3) < (unsigned)g.Wl ? c.y : (short)0;
        }
        const s16x4 o = {a.x, a.y, c.x, c.y};
        return o;
    };
    auto cat8 = [](s16x4 lo, s16x4 hi) __attribute__((always_inline)) -> bf16x8 {
        const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](const int set) __attribute__((always_inline)) {
        Cur cb = cc;
        if constexpr (PRO) advance(cb);
        bf16x8 gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = cat8(gs[set][0][q], gs[set][1][q]);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            s16x4 xa = xs[set][0][t], xb = xs[set][1][t];
            if constexpr (PRO) {
                const int dh = t == 0 ? dh0 : t == 1 ? dh1 : dh2, dw = t == 0 ? dw0 : t == 1 ? dw1 : dw2;
                xa = transform(xa, cc.i, cc.j, dh, dw);
                xb = transform(xb, cb.i, cb.j, dh, dw);
            }
            const bf16x8 xv = cat8(xa, xb);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv, gv[q], acc[t][q], 0, 0, 0);
        }
        if (need_bias) {       // column sums of G: rows of ones x this wave's g-tile r
            const bf16x8 gsel = r == 0 ? gv[0] : r == 1 ? gv[1] : r == 2 ? gv[2] : gv[3];
            bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), gsel, bacc, 0, 0, 0);
        }
        if constexpr (PRO) { cc = cb; advance(cc); }
    };

    // ---- the ring.  Step k (k = 0 .. nstages): my DMA of stage k has landed and my reads of stage k - 1 have retired (waitcnt),
    // everyone's (barrier); stage k + D - 1 goes into the slot stage k - 1 occupied; the fragments of stage k are requested into
    // one register set while the MFMAs of stage k - 1 run from the other.  A staging wave issues exactly four DMA instructions per
    // step (groups beyond the range and padding carry out-of-range offsets), the others none: the vmcnt count is exact for both.
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s);
    for (int k0 = 0; k0 <= nstages; k0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int k = k0 + u;
            if (k > nstages) break;
            wait_vm_lgkm0<(D - 2) * K::IPW>();
            __builtin_amdgcn_s_barrier();
            issue((u + D - 1) % D);
            read_stage(u, u & 1);
            if (k > 0) compute((u + 1) & 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // trailing (dead) stages: nothing may be in flight when the workgroup retires

    // ---- this wave's 12 tiles -> the workgroup's partial row [t][ci][co]; lane (b, j16) holds rows 4 b + e, column j16 of a tile
    {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.partial + (long)bxs * 3 * C * C, 0, 0xffffffffu, 0x00020000);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = cib * 64 + r * 16 + 4 * b + e, col = (4 * gh + q) * 16 + j16;
                    const float v = acc[t][q][e];      // (a named float: __builtin_bit_cast of the vector-element lvalue itself stores element 0 four times)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, ((t * C + row) * C + col) * 4, 0, 0);
                }
        if (need_bias && b == 0) a.bias_partial[(long)bxs * C + (4 * gh + r) * 16 + j16] = bacc[0];
    }
}

int g_ro_mode = 1;            // lf_debug_set_wgrad_ro: 0 = never (tapwgrad_kernel's job form), 1 = shipped
constexpr int RO_CAP64 = 512, RO_CAP128 = 256;      // workgroups per launch (the partial-row buffers are sized for these)
int g_ro_cap[2] = {RO_CAP64, RO_CAP128};            // ... A/B runs may lower them

bool ro_geom_ok(const LfTapGeom& g) {
    return g.ntaps == 3 && g.Cs == g.Cd && (g.Cs == 64 || g.Cs == 128) && g.ssh == 1 && g.ssw == 1 && g.dsh == 1 && g.dsw == 1 &&
           g.dah == 0 && g.daw == 0 && g.Hs == g.Hl && g.Ws == g.Wl && g.Hd == g.Hl && g.Wd == g.Wl && g.Wl % 16 == 0 &&
           g.s_pix % 8 == 0 && g.s_choff % 8 == 0 && g.d_pix % 8 == 0 && g.d_choff % 8 == 0 &&
           (long)g.N * g.Hs * g.Ws * g.s_pix * 2 < (long)LF_OOB && (long)g.N * g.Hd * g.Wd * g.d_pix * 2 < (long)LF_OOB;
}
// partial rows (= workgroups per x-block): at least 8 groups (128 pixels) per workgroup, a multiple of 8 (the kernel's per-XCD row
// dealing) when there are that many.  Every launch pays ~18 us that do not depend on its pixels -- ring fill, the partial rows'
// stores and the pass that sums them -- so a workgroup should see ~30+ stages: at 64 channels one workgroup per CU (256) until
// there are 64 groups for each of 512; at 128 channels 128 rows x 2 x-blocks = one 8-wave workgroup per CU (measured at batch 32
// 256 x 512 / batch 64 320 x 640, incl. the reduction: 64 channels 256 -> 24.9 / 52.3 us, 512 -> 28.5 / 45.1; 128 channels 64 rows
// 27.6 / 61.3, 128 rows 24.1 / 42.7).
int ro_rows(const LfTapGeom& g, int cap) {
    const long nrows = (long)g.N * g.Hl, gr = g.Wl / 16, ngroups = nrows * gr;
    long nsplit = cap / (g.Cs / 64);
    if (g.Cs == 64 && ngroups / 64 < nsplit) nsplit = ngroups / 64 > nsplit / 2 ? ngroups / 64 : nsplit / 2;
    const long rows_per_wg = (8 + gr - 1) / gr;
    const long most = (nrows + rows_per_wg - 1) / rows_per_wg;
    if (nsplit > most) nsplit = most;
    if (nsplit >= 8) nsplit &= ~7L;
    return nsplit < 1 ? 1 : (int)nsplit;
}

}  // namespace

// cap <= 0 restores the shipped cap (a lowered A/B cap must not stick to later production launches)
void lf_tapwgrad_ro_set(int mode, int cap64, int cap128) {
    g_ro_mode = mode;
    g_ro_cap[0] = cap64 > 0 ? (cap64 < RO_CAP64 ? cap64 : RO_CAP64) : RO_CAP64;
    g_ro_cap[1] = cap128 > 0 ? (cap128 < RO_CAP128 ? cap128 : RO_CAP128) : RO_CAP128;
}
// The kernels need 64-120 KB of dynamic LDS: hipFuncAttributeMaxDynamicSharedMemorySize once per device for all six instantiations,
// result cached under a mutex (host threads of one process may drive several GPUs).  A device that refuses takes the job form
// (tapwgrad_kernel) CONSISTENTLY: lf_tapwgrad_ro_ok() is what the launch, lf_tapwgrad_splits_for() and the sizing all ask.
// Without a current device (plan sizing on a host without GPU) the answer is "yes": the row BOUNDS do not depend on it.
static bool ro_lds_allowed() {
    static std::mutex mu;
    static std::map<int, bool> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find(dev);
    if (it != done.end()) return it->second;
    bool ok = true;
#define LF_RO_ATTR(CH, PROV) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(tapwgrad_ro_kernel<CH, PROV>), \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)RoCfg<CH>::LDS) == hipSuccess
    LF_RO_ATTR(64, 0); LF_RO_ATTR(64, 1); LF_RO_ATTR(64, 2); LF_RO_ATTR(128, 0); LF_RO_ATTR(128, 1); LF_RO_ATTR(128, 2);
#undef LF_RO_ATTR
    if (!ok) (void)hipGetLastError();
    return done[dev] = ok;
}
bool lf_tapwgrad_ro_ok(const LfTapGeom& g, int s16) { return g_ro_mode != 0 && s16 && ro_geom_ok(g) && ro_lds_allowed(); }
// partial rows the read-once kernel writes for this geometry at the SHIPPED caps (buffer sizing: independent of the A/B switches)
int lf_tapwgrad_ro_rows_bound(const LfTapGeom& g) { return ro_geom_ok(g) ? ro_rows(g, g.Cs == 128 ? RO_CAP128 : RO_CAP64) : 0; }
int lf_tapwgrad_ro_rows(const LfTapGeom& g) { return ro_rows(g, g_ro_cap[g.Cs == 128]); }

int lf_tapwgrad_ro_launch(const LfTapGeom& g, const LfWgradArgs& a, int pro, hipStream_t st) {
    LF_REQUIRE(a.s16 && ro_geom_ok(g), "tapwgrad_ro: geometry not supported");
    const int nsplit = lf_tapwgrad_ro_rows(g);
    LF_REQUIRE(nsplit <= lf_tapwgrad_ro_rows_bound(g), "tapwgrad_ro: %d partial rows exceed the sized %d", nsplit, lf_tapwgrad_ro_rows_bound(g));
    const int wb = (a.bias_partial != nullptr) | (g_ro_mode & 6);
    const dim3 grid((unsigned)(nsplit * (g.Cs / 64)));
    LF_REQUIRE(ro_lds_allowed(), "tapwgrad_ro: the device refuses the kernels' dynamic LDS (lf_tapwgrad_ro_ok() says so: take the job form)");
#define LF_RO(CH, PROV, SLOT)  /* PROV: see the kernel */                                                                                          \
    do {                                                                                                               \
        hipLaunchKernelGGL((tapwgrad_ro_kernel<CH, PROV>), grid, dim3(CH * 4), RoCfg<CH>::LDS, st, g, a, wb);          \
    } while (0)
    const bool horiz = g.tdw[0] != 0 || g.tdw[1] != 0 || g.tdw[2] != 0;
    if (g.Cs == 64) { if (pro != LF_PRO_BNRELU) LF_RO(64, 0, 0); else if (!horiz) LF_RO(64, 1, 1); else LF_RO(64, 2, 2); }
    else { if (pro != LF_PRO_BNRELU) LF_RO(128, 0, 3); else if (!horiz) LF_RO(128, 1, 4); else LF_RO(128, 2, 5); }
#undef LF_RO
    LF_CHECK_LAUNCH("tapwgrad_ro");
    return 0;
}
