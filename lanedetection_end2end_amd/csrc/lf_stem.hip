// The two ends of the backbone on the matrix cores: the 3-channel stem (DownsamplerBlock(3, 16) on the NCHW image: conv 3x3 s2 ||
// max-pool, BEV/Networks/ERFNet.py:11-22,66) forward and weight gradient, and the weight gradient of the 2x2 transposed-conv head
// (ERFNet.py:124,140).  Contract in lf_eltwise.h.
//
// These three launches were scalar-FMA kernels through round 4 -- 27 patch loads and 351 FMAs with LDS-broadcast weights per pixel
// in the stem, four passes over the patch in its weight gradient -- and sat 3-6x above what their bytes take: 72 / 120 / 57 us at
// batch 32, 256 x 512 where 117 / 117 / 100 MB need ~20 us (185 / 266 / 174 us at config 3's batch 64, 320 x 640).  They are small
// GEMMs with K = 27 (stem) or K = pixels (weight gradients): v_mfma_f32_16x16x4_f32 (exact fp32), operands by 4-byte buffer loads --
// the NHWC side contiguous 256 bytes per instruction, the NCHW image / logit-gradient side gathers that stay inside a few cache
// lines per instruction -- with the padding as out-of-range offsets (the load returns 0.0f).
#include "lf_eltwise.h"
#include "lf_types.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0xffff0000u;        // beyond every tensor the launchers admit
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)(bytes < (long)OOB ? bytes : (long)OOB), 0x00020000);
}
__device__ __forceinline__ float ldf(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// one element of an activation tensor (fp32, or bf16 widened exactly); `elem` = element index, OOB = zero
template <typename T>
__device__ __forceinline__ float lde(__amdgpu_buffer_rsrc_t r, unsigned elem, bool in) {
    if constexpr (sizeof(T) == 2) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, (int)(in ? elem * 2u : OOB), 0, 0) << 16);
    else return ldf(r, in ? elem * 4u : OOB);
}
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
// sum over the 16 lanes of a DPP row (lanes sharing l >> 4); every lane ends with the total
__device__ __forceinline__ float row_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// ---- stem forward.  D[co][pixel] = sum_k W[co][k] * patch[pixel][k], k = (ci, kh, kw): A = the weights (lane: row co = l & 15,
// k-slot l >> 4), B = the patches of a 16-pixel tile (lane: pixel l & 15, k-slot l >> 4), 7 K-steps for the 27 taps of 3 channels.
// A lane ends with channels 4 (l >> 4) .. + 3 of its pixel: one 16-byte store, 1 KB contiguous per instruction.  The pooled
// channels (Cc ..15: max over the 2x2 window of input channel co - Cc) take the place of the accumulator's padding rows.
// Workgroup = 256 pixels (one BatchNorm partial row, as before).
template <int CIN, typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ img, int N, int H, int W, const float* __restrict__ w,
                                                      const float* __restrict__ b, T* __restrict__ cat, float* __restrict__ rows, int ld) {
    constexpr int Cc = 16 - CIN, KK = 9 * CIN, KS = (KK + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 15, kq = lane >> 4;
    const int Ho = H / 2, Wo = W / 2;
    const unsigned npix = (unsigned)(N * Ho * Wo);
    const __amdgpu_buffer_rsrc_t ri = rsrc_of(img, (long)N * CIN * H * W * 4);
    float wa[KS];
    int toff[KS], tkh[KS], tkw[KS];
    bool tap[KS];                                                            // k < KK: a real tap (the last K-step is padded)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + kq, ci = k / 9, r = k - ci * 9, kh = r / 3, kw = r - kh * 3;
        tap[s] = k < KK;
        wa[s] = (tap[s] && pl < Cc) ? w[pl * KK + k] : 0.f;
        toff[s] = ci * H * W + (kh - 1) * W + (kw - 1);
        tkh[s] = kh; tkw[s] = kw;
    }
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = 4 * kq + e < Cc ? b[4 * kq + e] : 0.f;
    f32x4 s1 = zero4(), s2 = zero4();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned p = (unsigned)blockIdx.x * 256u + (unsigned)(wave * 64 + m * 16 + pl);
        const bool valid = p < npix;
        const unsigned q = valid ? p : 0u, rr = q / (unsigned)Wo;
        const int ow = (int)(q - rr * (unsigned)Wo), n = (int)(rr / (unsigned)Ho), oh = (int)(rr - (unsigned)n * (unsigned)Ho);
        const int base = (n * CIN * H + 2 * oh) * W + 2 * ow;                 // (ci = 0, ih = 2 oh, iw = 2 ow)
        f32x4 acc = zero4();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool in = valid && tap[s] && (oh > 0 || tkh[s] > 0) && (ow > 0 || tkw[s] > 0);    // (the far edges never pad: H, W even)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], ldf(ri, in ? (unsigned)(base + toff[s]) * 4u : OOB), acc, 0, 0, 0);
        }
        // pooled channels (round 6): lane (pixel pl, kq < CIN) takes the 2x2 maximum of input channel kq -- FOUR load instructions per
        // tile; the lanes kq = 3, which store the pooled channels 16 - CIN .. 15, fetch them with CIN lane permutes.  (Every lane used to
        // issue the 16 loads of its four output channels, 13 of them out-of-range dummies for most lanes: the kernel was bound by
        // the address path, 23 load instructions per 16 pixels.)
        float pm;
        {
            const bool pool = valid && kq < CIN;
            const unsigned o = (unsigned)(base + kq * H * W) * 4u;
            const float v0 = ldf(ri, pool ? o : OOB), v1 = ldf(ri, pool ? o + 4u : OOB), v2 = ldf(ri, pool ? o + (unsigned)W * 4u : OOB),
                        v3 = ldf(ri, pool ? o + (unsigned)W * 4u + 4u : OOB);
            pm = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
        }
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = 4 * kq + e;
            float pooled = 0.f;
            if (e >= 4 - CIN) pooled = __shfl(pm, (e - (4 - CIN)) * 16 + pl, 64);      // (compile-time e: CIN permutes, every lane takes part)
            out[e] = co < Cc ? acc[e] + bv[e] : pooled;
        }
        if (valid) lf_stv(cat + (size_t)p * 16 + 4 * kq, out);
        if (!valid) out = zero4();
        s1 += out; s2 += out * out;
    }
    if (rows) {
        __shared__ float red[4][2][16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = row_sum(s1[e]), c = row_sum(s2[e]);
            if (pl == 0) { red[wave][0][4 * kq + e] = a; red[wave][1][4 * kq + e] = c; }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int which = threadIdx.x >> 4, c = threadIdx.x & 15;
            rows[(long)(which * 16 + c) * ld + blockIdx.x] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);      // channel-major
        }
    }
}

// ---- stem weight gradient.  D[co][j] = sum_p G[p][co] * patch[p][j]: A = the gradient of the concat buffer (lane: channel l & 15 --
// the pooled channels >= Cc read as zero -- pixel slot l >> 4: 256 contiguous bytes per instruction), B = the patch column j
// (tiles of 16 columns; column KK is all ones: its row of D is the bias gradient), K = 4 pixels per MFMA, 16 pixels (one row
// segment: Wo % 16 == 0) per loop trip.  A wave walks the groups wave, wave + 4, ... of its workgroup's pixel range; the four
// waves are summed through LDS and the workgroup writes one partial row wrows[blk][co][j], brows[blk][co].
template <int CIN, typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ gcat, int N, int H, int W,
                                                        float* __restrict__ wrows, float* __restrict__ brows, int groups_per_wg) {
    constexpr int Cc = 16 - CIN, KK = 9 * CIN, NTL = (KK + 1 + 15) / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), pl = lane & 15, kq = lane >> 4;
    const int Ho = H / 2, Wo = W / 2;
    const unsigned ngroups = (unsigned)(N * Ho * Wo) >> 4, GR = (unsigned)Wo >> 4;
    const __amdgpu_buffer_rsrc_t ri = rsrc_of(img, (long)N * CIN * H * W * 4), rg = rsrc_of(gcat, (long)N * Ho * Wo * 16 * (long)sizeof(T));
    int joff[NTL], jkh[NTL], jkw[NTL], jkind[NTL];   // column j = 16 jt + pl: kind 0 = tap j, 1 = the ones column, 2 = padding
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
        const int j = 16 * jt + pl, ci = j / 9, r = j - ci * 9, kh = r / 3, kw = r - kh * 3;
        jkind[jt] = j < KK ? 0 : (j == KK ? 1 : 2);
        joff[jt] = ci * H * W + (kh - 1) * W + (kw - 1);
        jkh[jt] = kh; jkw[jt] = kw;
    }
    f32x4 acc[NTL];
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) acc[jt] = zero4();
    const unsigned g_lo = (unsigned)blockIdx.x * (unsigned)groups_per_wg;
    unsigned g_hi = g_lo + (unsigned)groups_per_wg;
    if (g_hi > ngroups) g_hi = ngroups;
    for (unsigned gi = g_lo + (unsigned)wave; gi < g_hi; gi += 4u) {
        const unsigned rr = gi / GR;                                             // wave-uniform
        const int ow0 = (int)(gi - rr * GR) * 16, n = (int)(rr / (unsigned)Ho), oh = (int)(rr - (unsigned)n * (unsigned)Ho);
        const unsigned gbase = (gi * 16u + (unsigned)kq) * 16u + (unsigned)pl;    // element of G: pixel gi * 16 + kq (+ 4 u), channel pl
        const int ibase = (n * CIN * H + 2 * oh) * W + 2 * (ow0 + kq);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a = lde<T>(rg, gbase + (unsigned)(64 * u), pl < Cc);
            const bool left = ow0 + kq + 4 * u > 0;
#pragma unroll
            for (int jt = 0; jt < NTL; ++jt) {
                const bool in = jkind[jt] == 0 && (oh > 0 || jkh[jt] > 0) && (left || jkw[jt] > 0);
                float x = ldf(ri, in ? (unsigned)(ibase + 8 * u + joff[jt]) * 4u : OOB);
                if (jkind[jt] == 1) x = 1.f;
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x, acc[jt], 0, 0, 0);
            }
        }
    }
    __shared__ float red[3][NTL * 4][64];
    if (wave > 0) {
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][jt * 4 + e][lane] = acc[jt][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[jt][e] + ((red[0][jt * 4 + e][lane] + red[1][jt * 4 + e][lane]) + red[2][jt * 4 + e][lane]);
                const int co = 4 * kq + e, j = 16 * jt + pl;                       // D[co][j]
                if (co < Cc && j < KK) wrows[((long)blockIdx.x * Cc + co) * KK + j] = v;
                if (co < Cc && j == KK && brows) brows[(long)blockIdx.x * Cc + co] = v;
            }
    }
}

// ---- head weight gradient.  ConvTranspose2d(16, K, 2, stride 2), weight (16, K, 2, 2): dW[ci][k][a][b] = sum_p x[p][ci] *
// gout[n][k][2 i + a][2 j + b], db[k] = sum gout.  D[ci][col] with col = k * 4 + a * 2 + b: A = x (lane: channel l & 15, pixel slot
// l >> 4: 256 contiguous bytes per instruction), B = the logit gradients of the pixel's 2x2 output window (gathers from the NCHW
// planes: the lanes of a column walk consecutive output pixels), K = 4 pixels per MFMA.  The bias gradient is the column sum of B,
// kept per lane and summed at the end.
template <int K, typename T>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ gout, float* __restrict__ wrows,
                                                        float* __restrict__ brows, int N, int h, int wd, int groups_per_wg) {
    constexpr int NC = K * 4, NTL = (NC + 15) / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), pl = lane & 15, kq = lane >> 4;
    const unsigned ngroups = (unsigned)(N * h * wd) >> 4, GR = (unsigned)wd >> 4;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(x, (long)N * h * wd * 16 * (long)sizeof(T)), rg = rsrc_of(gout, (long)N * K * 4 * h * wd * 4);
    int coff[NTL];                                                                 // column -> (plane k, a, b) element offset, -1 = none
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
        const int c = 16 * jt + pl, k = c >> 2, a = (c >> 1) & 1, b = c & 1;
        coff[jt] = c < NC ? (k * 2 * h + a) * (2 * wd) + b : -1;
    }
    f32x4 acc[NTL];
    float bsum[NTL];
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) { acc[jt] = zero4(); bsum[jt] = 0.f; }
    const unsigned g_lo = (unsigned)blockIdx.x * (unsigned)groups_per_wg;
    unsigned g_hi = g_lo + (unsigned)groups_per_wg;
    if (g_hi > ngroups) g_hi = ngroups;
    for (unsigned gi = g_lo + (unsigned)wave; gi < g_hi; gi += 4u) {
        const unsigned rr = gi / GR;
        const int j0 = (int)(gi - rr * GR) * 16, n = (int)(rr / (unsigned)h), i = (int)(rr - (unsigned)n * (unsigned)h);
        const unsigned xbase = (gi * 16u + (unsigned)kq) * 16u + (unsigned)pl;
        const int gbase = (n * K * 2 * h + 2 * i) * (2 * wd) + 2 * (j0 + kq);       // (k = 0, row 2 i, column 2 (j0 + kq))
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a = lde<T>(rx, xbase + (unsigned)(64 * u), true);
#pragma unroll
            for (int jt = 0; jt < NTL; ++jt) {
                const float gv = ldf(rg, coff[jt] >= 0 ? (unsigned)(gbase + 8 * u + coff[jt]) * 4u : OOB);
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gv, acc[jt], 0, 0, 0);
                bsum[jt] += gv;
            }
        }
    }
    __shared__ float red[3][NTL * 4][64];
    __shared__ float bred[4][NTL][64];
    if (wave > 0) {
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][jt * 4 + e][lane] = acc[jt][e];
    }
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) bred[wave][jt][lane] = bsum[jt];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[jt][e] + ((red[0][jt * 4 + e][lane] + red[1][jt * 4 + e][lane]) + red[2][jt * 4 + e][lane]);
                const int ci = 4 * kq + e, c = 16 * jt + pl;
                if (c < NC) wrows[((long)blockIdx.x * 16 + ci) * NC + c] = v;
            }
            // bias: column c summed over the four pixel slots and waves, then over the window (a, b) of class k
            float s = (bred[0][jt][lane] + bred[1][jt][lane]) + (bred[2][jt][lane] + bred[3][jt][lane]);
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            const int c = 16 * jt + pl;
            if (kq == 0 && (c & 3) == 0 && c < NC && brows) brows[(long)blockIdx.x * K + (c >> 2)] = s;
        }
    }
}

template <typename T> inline const T* as(const float* p) { return reinterpret_cast<const T*>(p); }
template <typename T> inline T* as(float* p) { return reinterpret_cast<T*>(p); }

// workgroups of the two weight gradients: one partial row each; at least 8 groups (128 pixels) per workgroup, at most 1024
inline int wgrad_wgs(long ngroups) {
    long g = (ngroups + 7) / 8;
    if (g > 1024) g = 1024;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

int lf_stem_rows(int N, int H, int W) { return lf_cdiv((long)N * (H / 2) * (W / 2), 256); }

int lf_stem_fwd(const float* img, int N, int Cin, int H, int W, const float* w, const float* b, float* cat, float* rows, int ld,
                int s16, hipStream_t st) {
    LF_REQUIRE(Cin >= 1 && Cin <= 4, "stem: in_channels %d not in 1..4", Cin);
    LF_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: odd image size");
    LF_REQUIRE((long)N * Cin * H * W * 4 < (long)OOB, "stem: image too large for 32-bit byte offsets");
    const dim3 grid(lf_stem_rows(N, H, W));
#define LF_STEM(CI)                                                                                                                  \
    do {                                                                                                                             \
        if (s16) hipLaunchKernelGGL((stem_fwd_kernel<CI, lf_bf16>), grid, dim3(256), 0, st, img, N, H, W, w, b, as<lf_bf16>(cat), rows, ld); \
        else hipLaunchKernelGGL((stem_fwd_kernel<CI, float>), grid, dim3(256), 0, st, img, N, H, W, w, b, cat, rows, ld);            \
    } while (0)
    switch (Cin) { case 1: LF_STEM(1); break; case 2: LF_STEM(2); break; case 3: LF_STEM(3); break; default: LF_STEM(4); break; }
#undef LF_STEM
    LF_CHECK_LAUNCH("stem_fwd");
    return 0;
}

int lf_stem_wgrad_rows(int N, int H, int W) { return wgrad_wgs((long)N * (H / 2) * (W / 2) / 16); }

int lf_stem_wgrad(const float* img, const float* gcat, int N, int Cin, int H, int W, float* wrows, float* brows,
                  int s16, hipStream_t st) {
    LF_REQUIRE(Cin >= 1 && Cin <= 4 && H % 2 == 0 && (W / 2) % 16 == 0, "stem_wgrad: in_channels %d / width %d not supported", Cin, W);
    LF_REQUIRE((long)N * Cin * H * W * 4 < (long)OOB && (long)N * (H / 2) * (W / 2) * 64 < (long)OOB, "stem_wgrad: tensor too large for 32-bit byte offsets");
    const long ngroups = (long)N * (H / 2) * (W / 2) / 16;
    const int wgs = lf_stem_wgrad_rows(N, H, W), gpw = (int)((ngroups + wgs - 1) / wgs);
#define LF_STEMW(CI)                                                                                                                 \
    do {                                                                                                                             \
        if (s16) hipLaunchKernelGGL((stem_wgrad_kernel<CI, lf_bf16>), dim3(wgs), dim3(256), 0, st, img, as<lf_bf16>(gcat), N, H, W, wrows, brows, gpw); \
        else hipLaunchKernelGGL((stem_wgrad_kernel<CI, float>), dim3(wgs), dim3(256), 0, st, img, gcat, N, H, W, wrows, brows, gpw); \
    } while (0)
    switch (Cin) { case 1: LF_STEMW(1); break; case 2: LF_STEMW(2); break; case 3: LF_STEMW(3); break; default: LF_STEMW(4); break; }
#undef LF_STEMW
    LF_CHECK_LAUNCH("stem_wgrad");
    return 0;
}

int lf_head_wgrad_rows(int N, int h, int w_) { return wgrad_wgs((long)N * h * w_ / 16); }

int lf_head_wgrad(const float* x, const float* gout, float* wrows, float* brows, int N, int h, int w_, int K,
                  int s16, hipStream_t st) {
    LF_REQUIRE(w_ % 16 == 0, "head_wgrad: width %d must be a multiple of 16", w_);
    LF_REQUIRE((long)N * K * 4 * h * w_ * 4 < (long)OOB && (long)N * h * w_ * 64 < (long)OOB, "head_wgrad: tensor too large for 32-bit byte offsets");
    const long ngroups = (long)N * h * w_ / 16;
    const int wgs = lf_head_wgrad_rows(N, h, w_), gpw = (int)((ngroups + wgs - 1) / wgs);
#define LF_HW(KK)                                                                                                                    \
    do {                                                                                                                             \
        if (s16) hipLaunchKernelGGL((head_wgrad_kernel<KK, lf_bf16>), dim3(wgs), dim3(256), 0, st, as<lf_bf16>(x), gout, wrows, brows, N, h, w_, gpw); \
        else hipLaunchKernelGGL((head_wgrad_kernel<KK, float>), dim3(wgs), dim3(256), 0, st, x, gout, wrows, brows, N, h, w_, gpw);   \
    } while (0)
    switch (K) {
        case 1: LF_HW(1); break; case 2: LF_HW(2); break; case 3: LF_HW(3); break; case 4: LF_HW(4); break; case 5: LF_HW(5); break;
        default: return lf_fail("head_wgrad: out_channels %d not in 1..5", K);
    }
#undef LF_HW
    LF_CHECK_LAUNCH("head_wgrad");
    return 0;
}
