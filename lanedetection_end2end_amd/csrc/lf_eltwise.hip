// Bandwidth-bound backbone kernels (contract in lf_eltwise.h).  All tensors NHWC fp32 unless
// noted; every global access is a 16-byte vector over the channel axis.
// Reference behaviour: BEV/Networks/ERFNet.py (DownsamplerBlock :11-22, non_bottleneck_1d :44-60,
// UpsamplerBlock :98-107, Decoder.output_conv :124), nn.BatchNorm2d(eps=1e-3), nn.Dropout2d.
#include "lf_eltwise.h"
#include "lf_types.h"
#include "lf_plan.h"      // LF_TRY, lf_rows_reduce_launch (lf_conv.h)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 z4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); return v;
}
__device__ __forceinline__ f32x4 pos4(f32x4 v, f32x4 m) {
    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f; return v;
}

// Reduce per-thread (a1,a2) over all threads sharing the same channel group (tid % U); thread tid < U
// ends up with the block totals.  256 threads, U a power of two dividing 256; NQ channel quads per thread.
template <int NQ>
__device__ __forceinline__ void block_reduce_quads(f32x4 (&a1)[NQ], f32x4 (&a2)[NQ], int U, float (*sm)[8 * NQ]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        sm[tid][8 * q + 0] = a1[q].x; sm[tid][8 * q + 1] = a1[q].y; sm[tid][8 * q + 2] = a1[q].z; sm[tid][8 * q + 3] = a1[q].w;
        sm[tid][8 * q + 4] = a2[q].x; sm[tid][8 * q + 5] = a2[q].y; sm[tid][8 * q + 6] = a2[q].z; sm[tid][8 * q + 7] = a2[q].w;
    }
    __syncthreads();
    for (int s = 128; s >= U; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int j = 0; j < 8 * NQ; ++j) sm[tid][j] += sm[tid + s][j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        a1[q].x = sm[tid][8 * q + 0]; a1[q].y = sm[tid][8 * q + 1]; a1[q].z = sm[tid][8 * q + 2]; a1[q].w = sm[tid][8 * q + 3];
        a2[q].x = sm[tid][8 * q + 4]; a2[q].y = sm[tid][8 * q + 5]; a2[q].z = sm[tid][8 * q + 6]; a2[q].w = sm[tid][8 * q + 7];
    }
}

struct StatParts { LfStatPart p[2]; int n; };

// Column sums of the partial rows of ONE channel by one FIN_THREADS-thread block.  The rows are stored CHANNEL-MAJOR (round 6):
// [2][C][ld] -- a channel's partial sums are one contiguous run of nrows floats per kind -- so a thread takes four consecutive rows
// of both kinds as two 16-byte loads, two such pairs in flight (2048 rows per trip), fp64 accumulation, wave shuffle + 4-slot LDS
// combine, fixed order.  Through round 5 the layout was [rows][2][C] and a block took 4 channels = 16 bytes of every 256- /
// 512-byte row half: every 128-byte line of the rows was fetched by 8 different blocks (16-32 blocks per launch), 7 us per launch at
// batch 32 and 11-12 us at config 3's 3200 rows -- and more loads in flight made it slower, not faster (U = 16: 21.8 us).
// Centred rows (LfStatPart::tile_pix > 0): sum v^2 about the origin = M2_r + (sum v)_r^2 / n_r, formed here in fp64 (n_r from the
// launch geometry).  Floats between nrows and ld are never used (select, not multiply: they may hold anything).
constexpr int FIN_THREADS = 256;
__device__ __forceinline__ void stat_channel_sums(const StatParts& sp, int c, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int k = 0; k < sp.n; ++k) {
        const LfStatPart& q = sp.p[k];
        const int cc = c - q.ch_off;
        if (cc < 0 || cc >= q.C) continue;
        const float* p1 = q.rows + (long)cc * q.ld;
        const float* p2 = q.rows + ((long)q.C + cc) * q.ld;
        for (int r0 = 4 * (int)threadIdx.x; r0 < q.nrows; r0 += 8 * FIN_THREADS) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + u * 4 * FIN_THREADS < q.nrows ? r0 + u * 4 * FIN_THREADS : 0;
                a[u] = ld4(p1 + r);
                b[u] = ld4(p2 + r);
            }
            asm volatile("" ::: "memory");      // all four requests are out before the first add
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = r0 + u * 4 * FIN_THREADS + e;
                    const bool v = r < q.nrows;
                    double inv = 0.0;
                    if (q.tile_pix > 0) {
                        const long left = q.seg_pix - (long)(r % q.seg_rows) * q.tile_pix;
                        const long n = left < q.tile_pix ? left : q.tile_pix;
                        inv = n > 0 ? 1.0 / (double)n : 0.0;
                    }
                    const double av = (double)a[u][e];
                    s1 += v ? av : 0.0;
                    s2 += v ? (double)b[u][e] + av * av * inv : 0.0;
                }
        }
    }
    constexpr int NWV = FIN_THREADS / 64;
    __shared__ double sm[NWV][2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w][0] = s1; sm[w][1] = s2; }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < NWV; ++k) { t1 += sm[k][0]; t2 += sm[k][1]; }      // fixed order: deterministic
    s1 = t1; s2 = t2;
}

// one block per channel
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_fwd_kernel(StatParts sp, int C, double count,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ rmean, float* __restrict__ rvar,
                                                            float momentum, float eps, int training,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ asc, float* __restrict__ ash) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (training) stat_channel_sums(sp, c, s1, s2);
    if (threadIdx.x == 0 && c < C) {
        double mean, var;
        if (training) {
            mean = s1 / count;
            var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mean);
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
        } else {
            mean = (double)rmean[c];
            var = (double)rvar[c];
        }
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double sc = (double)gamma[c] * rstd;
        scale[c] = (float)sc;
        shift[c] = (float)((double)beta[c] - mean * sc);
        asc[c] = (float)rstd;
        ash[c] = (float)(-mean * rstd);
    }
}

// NQ channel quads per thread (lf_ldq: 16 bytes per lane for either storage type); U = threads per pixel = C / (4 NQ)
template <typename T, int NQ>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ x, const float* __restrict__ sc,
                                                    const float* __restrict__ sh, const float* __restrict__ dm,
                                                    const T* __restrict__ res, T* __restrict__ y, long units,
                                                    int U, long pix_per_image) {
    constexpr int V = 4 * NQ;
    const int C = U * V;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
        const int c = (int)(u % U) * V;
        f32x4 v[NQ];
        lf_ldq<T, NQ>(x + u * V, v);
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = v[q] * ld4(sc + c + 4 * q) + ld4(sh + c + 4 * q);
        if (dm) {
            const float* d = dm + ((u / U) / pix_per_image) * C + c;
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[q] *= ld4(d + 4 * q);
        }
        if (res) {
            f32x4 r[NQ];
            lf_ldq<T, NQ>(res + u * V, r);
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[q] += r[q];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = relu4(v[q]);
        lf_stq<T, NQ>(y + u * V, v);
    }
}

template <typename T, int NQ>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                           const T* __restrict__ t, const float* __restrict__ asc,
                                                           const float* __restrict__ ash, const float* __restrict__ dm,
                                                           float* __restrict__ rows, int ld, long npix, int U, long pix_per_image) {
    constexpr int V = 4 * NQ;
    const int C = U * V, ppi = 256 / U;
    const int cq = threadIdx.x % U, pr = threadIdx.x / U;
    const int c = cq * V;
    (void)asc; (void)ash;
    f32x4 a1[NQ], a2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { a1[q] = z4(); a2[q] = z4(); }
    const long per_block = ((npix + gridDim.x - 1) / gridDim.x + ppi - 1) / ppi * ppi;
    const long p0 = (long)blockIdx.x * per_block;
    long p1 = p0 + per_block;
    if (p1 > npix) p1 = npix;
    for (long p = p0 + pr; p < p1; p += ppi) {
        const long off = p * C + c;
        f32x4 gm[NQ], tv[NQ];
        lf_ldq<T, NQ>(g + off, gm);
        lf_ldq<T, NQ>(t + off, tv);
        if (y) {
            f32x4 yv[NQ];
            lf_ldq<T, NQ>(y + off, yv);
#pragma unroll
            for (int q = 0; q < NQ; ++q) gm[q] = pos4(gm[q], yv[q]);
        }
        if (dm) {
            const float* d = dm + (p / pix_per_image) * C + c;
#pragma unroll
            for (int q = 0; q < NQ; ++q) gm[q] *= ld4(d + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            a1[q] += gm[q];
            a2[q] += gm[q] * tv[q];          // RAW: bn_bwd_finalize_kernel applies the normalisation in fp64
        }
    }
    __shared__ float sm[256][8 * NQ];
    block_reduce_quads<NQ>(a1, a2, U, sm);
    if (threadIdx.x < U) {         // channel-major rows: [2][C][ld]
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rows[(long)(c + 4 * q + e) * ld + blockIdx.x] = a1[q][e];
                rows[((long)C + c + 4 * q + e) * ld + blockIdx.x] = a2[q][e];
            }
    }
}

// training = 0 (the forward normalised with the running statistics): BatchNorm is a per-channel affine map, the mean
// terms of the data gradient vanish (c1 = c2 = 0); the parameter gradients are the same sums.
// (mean_scale = 1 / count in training mode, 0 in eval mode -- decided on the host: a run-time `training ? :` in here made hipcc
// unroll the final sum into 128 registers + 564 bytes of scratch, 19 us per launch instead of 6)
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(StatParts sp, int C, double mean_scale, const float* __restrict__ asc,
                                                            const float* __restrict__ ash, float* __restrict__ c1,
                                                            float* __restrict__ c2, float* __restrict__ ggamma,
                                                            float* __restrict__ gbeta) {
    // rows: [sum g, sum g * t] with t the PRE-BatchNorm tensor (raw: the producing epilogues carry no per-channel vectors);
    // sum g * x^ = rstd * sum g t - mean rstd * sum g, formed here in fp64 from the fp64 column sums
    const int c = blockIdx.x;
    double t1, t2;
    stat_channel_sums(sp, c, t1, t2);
    if (threadIdx.x == 0 && c < C) {
        const double rstd = (double)asc[c], mr = (double)ash[c];       // mr = -mean * rstd
        t2 = rstd * t2 + mr * t1;
        c1[c] = (float)(t1 * mean_scale);
        c2[c] = (float)(t2 * mean_scale);
        ggamma[c] = (float)t2;
        gbeta[c] = (float)t1;
    }
}

template <typename T, int NQ>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                          const T* __restrict__ t, const float* __restrict__ asc,
                                                          const float* __restrict__ ash, const float* __restrict__ gamma,
                                                          const float* __restrict__ c1, const float* __restrict__ c2,
                                                          const float* __restrict__ dm, T* __restrict__ g_t,
                                                          T* __restrict__ g_z, long units, int U, long pix_per_image) {
    constexpr int V = 4 * NQ;
    const int C = U * V;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
        const int c = (int)(u % U) * V;
        f32x4 gm[NQ], tv[NQ];
        lf_ldq<T, NQ>(g + u * V, gm);
        lf_ldq<T, NQ>(t + u * V, tv);
        if (y) {
            f32x4 yv[NQ];
            lf_ldq<T, NQ>(y + u * V, yv);
#pragma unroll
            for (int q = 0; q < NQ; ++q) gm[q] = pos4(gm[q], yv[q]);
        }
        if (g_z) lf_stq<T, NQ>(g_z + u * V, gm);
        if (dm) {
            const float* d = dm + ((u / U) / pix_per_image) * C + c;
#pragma unroll
            for (int q = 0; q < NQ; ++q) gm[q] *= ld4(d + 4 * q);
        }
        f32x4 o[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 rstd = ld4(asc + c + 4 * q);
            const f32x4 xh = tv[q] * rstd + ld4(ash + c + 4 * q);
            o[q] = ld4(gamma + c + 4 * q) * rstd * (gm[q] - ld4(c1 + c + 4 * q) - xh * ld4(c2 + c + 4 * q));
        }
        lf_stq<T, NQ>(g_t + u * V, o);
    }
}

// ---- max-pool branch of DownsamplerBlock ------------------------------------------------
template <typename T, int NQ>
__global__ __launch_bounds__(256) void pool_concat_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int U,
                                                             T* __restrict__ cat, int cat_pix, int choff,
                                                             float* __restrict__ rows, int ld) {
    constexpr int V = 4 * NQ;
    const int C = U * V, ppi = 256 / U, Ho = H / 2, Wo = W / 2;
    const int cq = threadIdx.x % U, pr = threadIdx.x / U, c = cq * V;
    const long npix = (long)N * Ho * Wo;
    const long per_block = ((npix + gridDim.x - 1) / gridDim.x + ppi - 1) / ppi * ppi;
    const long p0 = (long)blockIdx.x * per_block;
    long p1 = p0 + per_block;
    if (p1 > npix) p1 = npix;
    f32x4 a1[NQ], a2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { a1[q] = z4(); a2[q] = z4(); }
    for (long p = p0 + pr; p < p1; p += ppi) {
        const int ow = (int)(p % Wo);
        const long r = p / Wo;
        const int oh = (int)(r % Ho), n = (int)(r / Ho);
        const T* b = x + (((long)n * H + 2 * oh) * W + 2 * ow) * C + c;
        f32x4 v[NQ], u1[NQ], u2[NQ], u3[NQ];
        lf_ldq<T, NQ>(b, v);
        lf_ldq<T, NQ>(b + C, u1);
        lf_ldq<T, NQ>(b + (long)W * C, u2);
        lf_ldq<T, NQ>(b + (long)W * C + C, u3);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = fmaxf(fmaxf(fmaxf(v[q][e], u1[q][e]), u2[q][e]), u3[q][e]);
            a1[q] += v[q];
            a2[q] += v[q] * v[q];
        }
        lf_stq<T, NQ>(cat + p * cat_pix + choff + c, v);
    }
    __shared__ float sm[256][8 * NQ];
    block_reduce_quads<NQ>(a1, a2, U, sm);
    if (rows && threadIdx.x < U) {         // channel-major rows: [2][C][ld]
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rows[(long)(c + 4 * q + e) * ld + blockIdx.x] = a1[q][e];
                rows[((long)C + c + 4 * q + e) * ld + blockIdx.x] = a2[q][e];
            }
    }
}

template <typename T, int NQ>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gcat, int N,
                                                      int H, int W, int U, int cat_pix, int choff, T* __restrict__ gx) {
    constexpr int V = 4 * NQ;
    const int C = U * V, Ho = H / 2, Wo = W / 2;
    const long units = (long)N * Ho * Wo * U;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
        const int c = (int)(u % U) * V;
        const long p = u / U;
        const int ow = (int)(p % Wo);
        const long r = p / Wo;
        const int oh = (int)(r % Ho), n = (int)(r / Ho);
        const long base = (((long)n * H + 2 * oh) * W + 2 * ow) * C + c;
        const long o1 = C, o2 = (long)W * C, o3 = (long)W * C + C;
        f32x4 v0[NQ], v1[NQ], v2[NQ], v3[NQ], g[NQ], r0[NQ], r1[NQ], r2[NQ], r3[NQ];
        lf_ldq<T, NQ>(x + base, v0); lf_ldq<T, NQ>(x + base + o1, v1); lf_ldq<T, NQ>(x + base + o2, v2); lf_ldq<T, NQ>(x + base + o3, v3);
        lf_ldq<T, NQ>(gcat + p * cat_pix + choff + c, g);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // first maximum in window scan order wins (strict >), as ATen's max_pool2d
                int arg = 0;
                float m = v0[q][e];
                if (v1[q][e] > m) { m = v1[q][e]; arg = 1; }
                if (v2[q][e] > m) { m = v2[q][e]; arg = 2; }
                if (v3[q][e] > m) { m = v3[q][e]; arg = 3; }
                r0[q][e] = arg == 0 ? g[q][e] : 0.f; r1[q][e] = arg == 1 ? g[q][e] : 0.f;
                r2[q][e] = arg == 2 ? g[q][e] : 0.f; r3[q][e] = arg == 3 ? g[q][e] : 0.f;
            }
        lf_stq<T, NQ>(gx + base, r0); lf_stq<T, NQ>(gx + base + o1, r1); lf_stq<T, NQ>(gx + base + o2, r2); lf_stq<T, NQ>(gx + base + o3, r3);
    }
}

// ---- head: ConvTranspose2d(16, K, 2, stride=2), weight (16,K,2,2) -------------------------
constexpr int HEAD_MAXK = 8;
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ out, int N, int h,
                                                      int wd, int K) {
    __shared__ float sw[16 * HEAD_MAXK * 4 + HEAD_MAXK];
    for (int i = threadIdx.x; i < 16 * K * 4; i += 256) sw[i] = w[i];
    for (int i = threadIdx.x; i < K; i += 256) sw[16 * HEAD_MAXK * 4 + i] = b[i];
    __syncthreads();
    const long npix = (long)N * h * wd;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const int j = (int)(p % wd);
        const long r = p / wd;
        const int i = (int)(r % h), n = (int)(r / h);
        float xv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = lf_ldv(x + p * 16 + q * 4);
            xv[q * 4] = v.x; xv[q * 4 + 1] = v.y; xv[q * 4 + 2] = v.z; xv[q * 4 + 3] = v.w;
        }
        for (int k = 0; k < K; ++k) {
            float o[4];
            const float bk = sw[16 * HEAD_MAXK * 4 + k];
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) o[ab] = bk;
#pragma unroll
            for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) o[ab] = fmaf(xv[ci], sw[(ci * K + k) * 4 + ab], o[ab]);
            float* op = out + (((long)n * K + k) * (2 * h) + 2 * i) * (2 * wd) + 2 * j;
            *reinterpret_cast<float2*>(op) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2*>(op + 2 * wd) = make_float2(o[2], o[3]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_data_kernel(const float* __restrict__ gout, const float* __restrict__ w,
                                                           T* __restrict__ gx, int N, int h, int wd, int K) {
    __shared__ float sw[16 * HEAD_MAXK * 4];
    for (int i = threadIdx.x; i < 16 * K * 4; i += 256) sw[i] = w[i];
    __syncthreads();
    const long npix = (long)N * h * wd;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const int j = (int)(p % wd);
        const long r = p / wd;
        const int i = (int)(r % h), n = (int)(r / h);
        float acc[16];
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) acc[ci] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float* gp = gout + (((long)n * K + k) * (2 * h) + 2 * i) * (2 * wd) + 2 * j;
            const float2 g0 = *reinterpret_cast<const float2*>(gp);
            const float2 g1 = *reinterpret_cast<const float2*>(gp + 2 * wd);
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) {
                const float* wk = sw + (ci * K + k) * 4;
                acc[ci] = fmaf(g0.x, wk[0], fmaf(g0.y, wk[1], fmaf(g1.x, wk[2], fmaf(g1.y, wk[3], acc[ci]))));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
            lf_stv(gx + p * 16 + q * 4, v);
        }
    }
}

// the two big streaming passes (BN apply, BN-backward apply): one unit per thread up to 16384 workgroups -- A/B on one box at batch
// 32: caps 2048 / 4096 / 16384 -> 16.31 / 16.28 / 16.24 ms per step (the loads of one iteration are issued one after the other,
// so a thread that loops hides less latency than a fresh one)
constexpr int LF_STREAM_BLOCKS = 16384;
inline int grid_for(long units, int cap) {
    long g = (units + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
inline bool quad_ok(int C) { return C % 4 == 0 && 256 % (C / 4) == 0; }
// bf16 storage with 8-channel groups (16 bytes per lane): every offset the kernels form must be a multiple of 8 elements
inline bool oct_ok(int s16, int C) { return s16 && C % 8 == 0 && 256 % (C / 8) == 0; }
// activation pointers travel as float* through the host code; s16 says they hold bf16 elements
template <typename T> inline const T* as(const float* p) { return reinterpret_cast<const T*>(p); }
template <typename T> inline T* as(float* p) { return reinterpret_cast<T*>(p); }
#define LF_BY_STORAGE(s16, CALL_BF16, CALL_F32) do { if (s16) { CALL_BF16; } else { CALL_F32; } } while (0)
#define LF_BY_STORAGE3(s16, oct, CALL_BF16X8, CALL_BF16, CALL_F32) do { if (oct) { CALL_BF16X8; } else if (s16) { CALL_BF16; } else { CALL_F32; } } while (0)

}  // namespace

int lf_bn_finalize_fwd(const LfStatPart* parts, int nparts, int C, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, int training, float* scale,
                       float* shift, float* asc, float* ash, hipStream_t st) {
    LF_REQUIRE(nparts >= 0 && nparts <= 2, "bn_finalize: at most 2 partial sources");
    StatParts sp;
    sp.n = nparts;
    for (int i = 0; i < nparts; ++i) sp.p[i] = parts[i];
    for (int i = 0; i < nparts; ++i) LF_REQUIRE(parts[i].C % 4 == 0 && parts[i].ch_off % 4 == 0, "bn_finalize: channel ranges must be multiples of 4");
    for (int i = 0; i < nparts; ++i)
        LF_REQUIRE(parts[i].tile_pix == 0 || (parts[i].seg_rows > 0 && parts[i].seg_pix > 0 && parts[i].nrows % parts[i].seg_rows == 0),
                   "bn_finalize: centred rows need the launch geometry (seg_rows, seg_pix)");
    for (int i = 0; i < nparts; ++i)
        LF_REQUIRE(parts[i].ld >= parts[i].nrows && parts[i].ld % 4 == 0 && ((size_t)parts[i].rows & 15) == 0,
                   "bn_finalize: channel-major rows need a 16-byte aligned base and a leading dimension >= nrows, multiple of 4 (ld=%d nrows=%d)", parts[i].ld, parts[i].nrows);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(FIN_THREADS), 0, st, sp, C, count, gamma, beta,
                       running_mean, running_var, momentum, eps, training, scale, shift, asc, ash);
    LF_CHECK_LAUNCH("bn_finalize_fwd");
    return 0;
}

int lf_bn_act(const float* x, const float* sc, const float* sh, const float* dm, const float* res, float* y, long npix,
              int C, long pix_per_image, int s16, hipStream_t st) {
    LF_REQUIRE(C % 4 == 0, "bn_act: C %% 4");
    const bool oct = oct_ok(s16, C);
    const int V = oct ? 8 : 4;
    const long units = npix * (C / V);
    const dim3 grid(grid_for(units, LF_STREAM_BLOCKS));
    LF_BY_STORAGE3(s16, oct,
        hipLaunchKernelGGL((bn_act_kernel<lf_bf16, 2>), grid, dim3(256), 0, st, as<lf_bf16>(x), sc, sh, dm, as<lf_bf16>(res), as<lf_bf16>(y), units, C / V, pix_per_image),
        hipLaunchKernelGGL((bn_act_kernel<lf_bf16, 1>), grid, dim3(256), 0, st, as<lf_bf16>(x), sc, sh, dm, as<lf_bf16>(res), as<lf_bf16>(y), units, C / V, pix_per_image),
        hipLaunchKernelGGL((bn_act_kernel<float, 1>), grid, dim3(256), 0, st, x, sc, sh, dm, res, y, units, C / V, pix_per_image));
    LF_CHECK_LAUNCH("bn_act");
    return 0;
}

int lf_bn_bwd_reduce_rows(long npix) { return grid_for(npix * 4, 1024); }

int lf_bn_bwd_reduce(const float* g, const float* y, const float* t, const float* asc, const float* ash, const float* dm,
                     float* rows, int ld, long npix, int C, long pix_per_image, int s16, hipStream_t st) {
    LF_REQUIRE(quad_ok(C), "bn_bwd_reduce: unsupported channel count %d", C);
    const dim3 grid(lf_bn_bwd_reduce_rows(npix));
    const bool oct = oct_ok(s16, C);
    const int V = oct ? 8 : 4;
    LF_BY_STORAGE3(s16, oct,
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<lf_bf16, 2>), grid, dim3(256), 0, st, as<lf_bf16>(g), as<lf_bf16>(y), as<lf_bf16>(t), asc, ash, dm, rows, ld, npix, C / V, pix_per_image),
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<lf_bf16, 1>), grid, dim3(256), 0, st, as<lf_bf16>(g), as<lf_bf16>(y), as<lf_bf16>(t), asc, ash, dm, rows, ld, npix, C / V, pix_per_image),
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1>), grid, dim3(256), 0, st, g, y, t, asc, ash, dm, rows, ld, npix, C / V, pix_per_image));
    LF_CHECK_LAUNCH("bn_bwd_reduce");
    return 0;
}

int lf_bn_bwd_finalize(const LfStatPart* parts, int nparts, int C, double count, const float* asc, const float* ash, float* c1,
                       float* c2, float* ggamma, float* gbeta, int training, hipStream_t st) {
    LF_REQUIRE(nparts >= 1 && nparts <= 2, "bn_bwd_finalize: 1..2 partial sources");
    StatParts sp;
    sp.n = nparts;
    for (int i = 0; i < nparts; ++i) sp.p[i] = parts[i];
    for (int i = 0; i < nparts; ++i) LF_REQUIRE(parts[i].C % 4 == 0 && parts[i].ch_off % 4 == 0, "bn_bwd_finalize: channel ranges must be multiples of 4");
    for (int i = 0; i < nparts; ++i)
        LF_REQUIRE(parts[i].ld >= parts[i].nrows && parts[i].ld % 4 == 0 && ((size_t)parts[i].rows & 15) == 0,
                   "bn_bwd_finalize: channel-major rows need a 16-byte aligned base and a leading dimension >= nrows, multiple of 4 (ld=%d nrows=%d)", parts[i].ld, parts[i].nrows);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, sp, C, training ? 1.0 / count : 0.0, asc, ash,
                       c1, c2, ggamma, gbeta);
    LF_CHECK_LAUNCH("bn_bwd_finalize");
    return 0;
}

int lf_bn_bwd_apply(const float* g, const float* y, const float* t, const float* asc, const float* ash, const float* gamma,
                    const float* c1, const float* c2, const float* dm, float* g_t, float* g_z, long npix, int C,
                    long pix_per_image, int s16, hipStream_t st) {
    LF_REQUIRE(C % 4 == 0, "bn_bwd_apply: C %% 4");
    const bool oct = oct_ok(s16, C);
    const int V = oct ? 8 : 4;
    const long units = npix * (C / V);
    const dim3 grid(grid_for(units, LF_STREAM_BLOCKS));
    LF_BY_STORAGE3(s16, oct,
        hipLaunchKernelGGL((bn_bwd_apply_kernel<lf_bf16, 2>), grid, dim3(256), 0, st, as<lf_bf16>(g), as<lf_bf16>(y), as<lf_bf16>(t), asc, ash, gamma, c1, c2, dm, as<lf_bf16>(g_t), as<lf_bf16>(g_z), units, C / V, pix_per_image),
        hipLaunchKernelGGL((bn_bwd_apply_kernel<lf_bf16, 1>), grid, dim3(256), 0, st, as<lf_bf16>(g), as<lf_bf16>(y), as<lf_bf16>(t), asc, ash, gamma, c1, c2, dm, as<lf_bf16>(g_t), as<lf_bf16>(g_z), units, C / V, pix_per_image),
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 1>), grid, dim3(256), 0, st, g, y, t, asc, ash, gamma, c1, c2, dm, g_t, g_z, units, C / V, pix_per_image));
    LF_CHECK_LAUNCH("bn_bwd_apply");
    return 0;
}

int lf_pool_rows(long npix_out) { return grid_for(npix_out * 4, 1024); }

int lf_pool_concat_fwd(const float* x, int N, int H, int W, int Cin, float* cat, int cat_pix, int choff, float* rows, int ld,
                       int s16, hipStream_t st) {
    LF_REQUIRE(quad_ok(Cin) && H % 2 == 0 && W % 2 == 0, "pool_concat: unsupported shape");
    const long npo = (long)N * (H / 2) * (W / 2);
    const dim3 grid(lf_pool_rows(npo));
    const bool oct = oct_ok(s16, Cin) && cat_pix % 8 == 0 && choff % 8 == 0;
    const int V = oct ? 8 : 4;
    LF_BY_STORAGE3(s16, oct,
        hipLaunchKernelGGL((pool_concat_fwd_kernel<lf_bf16, 2>), grid, dim3(256), 0, st, as<lf_bf16>(x), N, H, W, Cin / V, as<lf_bf16>(cat), cat_pix, choff, rows, ld),
        hipLaunchKernelGGL((pool_concat_fwd_kernel<lf_bf16, 1>), grid, dim3(256), 0, st, as<lf_bf16>(x), N, H, W, Cin / V, as<lf_bf16>(cat), cat_pix, choff, rows, ld),
        hipLaunchKernelGGL((pool_concat_fwd_kernel<float, 1>), grid, dim3(256), 0, st, x, N, H, W, Cin / V, cat, cat_pix, choff, rows, ld));
    LF_CHECK_LAUNCH("pool_concat_fwd");
    return 0;
}

int lf_pool_bwd(const float* x, const float* gcat, int N, int H, int W, int Cin, int cat_pix, int choff, float* gx,
                int s16, hipStream_t st) {
    const bool oct = oct_ok(s16, Cin) && cat_pix % 8 == 0 && choff % 8 == 0;
    const int V = oct ? 8 : 4;
    const long units = (long)N * (H / 2) * (W / 2) * (Cin / V);
    const dim3 grid(grid_for(units, 4096));
    LF_BY_STORAGE3(s16, oct,
        hipLaunchKernelGGL((pool_bwd_kernel<lf_bf16, 2>), grid, dim3(256), 0, st, as<lf_bf16>(x), as<lf_bf16>(gcat), N, H, W, Cin / V, cat_pix, choff, as<lf_bf16>(gx)),
        hipLaunchKernelGGL((pool_bwd_kernel<lf_bf16, 1>), grid, dim3(256), 0, st, as<lf_bf16>(x), as<lf_bf16>(gcat), N, H, W, Cin / V, cat_pix, choff, as<lf_bf16>(gx)),
        hipLaunchKernelGGL((pool_bwd_kernel<float, 1>), grid, dim3(256), 0, st, x, gcat, N, H, W, Cin / V, cat_pix, choff, gx));
    LF_CHECK_LAUNCH("pool_bwd");
    return 0;
}

int lf_head_fwd(const float* x, const float* w, const float* b, float* out, int N, int h, int w_, int K, int s16,
                hipStream_t st) {
    LF_REQUIRE(K >= 1 && K <= HEAD_MAXK, "head: out_channels %d not in 1..%d", K, HEAD_MAXK);
    const dim3 grid(grid_for((long)N * h * w_, 4096));
    LF_BY_STORAGE(s16,
        hipLaunchKernelGGL(head_fwd_kernel<lf_bf16>, grid, dim3(256), 0, st, as<lf_bf16>(x), w, b, out, N, h, w_, K),
        hipLaunchKernelGGL(head_fwd_kernel<float>, grid, dim3(256), 0, st, x, w, b, out, N, h, w_, K));
    LF_CHECK_LAUNCH("head_fwd");
    return 0;
}

int lf_head_bwd_data(const float* gout, const float* w, float* gx, int N, int h, int w_, int K, int s16, hipStream_t st) {
    LF_REQUIRE(K >= 1 && K <= HEAD_MAXK, "head: out_channels %d not in 1..%d", K, HEAD_MAXK);
    const dim3 grid(grid_for((long)N * h * w_, 4096));
    LF_BY_STORAGE(s16,
        hipLaunchKernelGGL(head_bwd_data_kernel<lf_bf16>, grid, dim3(256), 0, st, gout, w, as<lf_bf16>(gx), N, h, w_, K),
        hipLaunchKernelGGL(head_bwd_data_kernel<float>, grid, dim3(256), 0, st, gout, w, gx, N, h, w_, K));
    LF_CHECK_LAUNCH("head_bwd_data");
    return 0;
}

// ---------------------------------------------------------------------------------------
// encoder.output_conv: Conv2d(128, K, 1) on the encoder output -- the `predict=True` branch of Encoder.forward that
// Net.forward(only_encode=True) returns (BEV/Networks/ERFNet.py:84,86-95,151-153).  NHWC (N,h,w,C) fp32 in, NCHW (N,K,h,w) out.
// Bandwidth-bound (reads C floats, writes K <= 8 per pixel): one thread per pixel, weights broadcast from LDS.
// ---------------------------------------------------------------------------------------
namespace {
constexpr int PW_MAXK = 8, PW_MAXC = 256;

__global__ __launch_bounds__(256) void pointwise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, long npix,
                                                           long pix_per_image, int C, int K) {
    __shared__ float sw[PW_MAXK * PW_MAXC];
    for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        float acc[PW_MAXK];
        for (int k = 0; k < K; ++k) acc[k] = b ? b[k] : 0.f;
        const float* xp = x + p * C;
        for (int c = 0; c < C; c += 4) {
            const lf_f32x4 v = *reinterpret_cast<const lf_f32x4*>(xp + c);
            for (int k = 0; k < K; ++k) {
                const float* wk = sw + k * C + c;
                acc[k] = fmaf(v.x, wk[0], fmaf(v.y, wk[1], fmaf(v.z, wk[2], fmaf(v.w, wk[3], acc[k]))));
            }
        }
        const long n = p / pix_per_image, q = p - n * pix_per_image;
        for (int k = 0; k < K; ++k) y[(n * K + k) * pix_per_image + q] = acc[k];
    }
}

// gx[p][c] = sum_k gy[n][k][q] * w[k][c]
__global__ __launch_bounds__(256) void pointwise_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                float* __restrict__ gx, long npix, long pix_per_image, int C,
                                                                int K) {
    __shared__ float sw[PW_MAXK * PW_MAXC];
    for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const long n = p / pix_per_image, q = p - n * pix_per_image;
        float g[PW_MAXK];
        for (int k = 0; k < K; ++k) g[k] = gy[(n * K + k) * pix_per_image + q];
        float* gp = gx + p * C;
        for (int c = 0; c < C; c += 4) {
            lf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; ++k) {
                const float* wk = sw + k * C + c;
                v.x = fmaf(g[k], wk[0], v.x); v.y = fmaf(g[k], wk[1], v.y); v.z = fmaf(g[k], wk[2], v.z); v.w = fmaf(g[k], wk[3], v.w);
            }
            *reinterpret_cast<lf_f32x4*>(gp + c) = v;
        }
    }
}

// partial rows: wrows[block][k][c] = sum over the block's pixels of gy[k] * x[c]; brows[block][k] = sum gy[k].
// thread = channel (C <= 256 threads active), pixels of the block in sequence: x loads coalesce over c, gy broadcasts.
__global__ __launch_bounds__(256) void pointwise_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             float* __restrict__ wrows, float* __restrict__ brows, long npix,
                                                             long pix_per_image, int C, int K, int pix_per_block) {
    const int c = threadIdx.x;
    const long p0 = (long)blockIdx.x * pix_per_block;
    float acc[PW_MAXK], bs[PW_MAXK];
    for (int k = 0; k < K; ++k) { acc[k] = 0.f; bs[k] = 0.f; }
    for (int i = 0; i < pix_per_block; ++i) {
        const long p = p0 + i;
        if (p >= npix) break;
        const long n = p / pix_per_image, q = p - n * pix_per_image;
        const float xv = c < C ? x[p * C + c] : 0.f;
        for (int k = 0; k < K; ++k) {
            const float g = gy[(n * K + k) * pix_per_image + q];
            acc[k] = fmaf(g, xv, acc[k]);
            bs[k] += g;
        }
    }
    if (c < C)
        for (int k = 0; k < K; ++k) wrows[((long)blockIdx.x * K + k) * C + c] = acc[k];
    if (c == 0)
        for (int k = 0; k < K; ++k) brows[(long)blockIdx.x * K + k] = bs[k];
}
}  // namespace

extern "C" {
// y (N,K,h,w) NCHW = Conv2d(C, K, 1)(x), x (N,h,w,C) NHWC fp32 (the encoder output in place), w (K,C) = the Conv2d weight
// (K,C,1,1) flattened, b (K) or NULL.  C % 4 == 0, C <= 256, K <= 8.
int lf_pointwise_fwd(const float* x, const float* w, const float* b, float* y, int N, int h, int w_, int C, int K, void* stream) {
    LF_REQUIRE(x && w && y && C % 4 == 0 && C <= PW_MAXC && K >= 1 && K <= PW_MAXK, "lf_pointwise_fwd: bad arguments (C=%d K=%d)", C, K);
    const long npix = (long)N * h * w_;
    hipLaunchKernelGGL(pointwise_fwd_kernel, dim3(grid_for(npix, 4096)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, npix,
                       (long)h * w_, C, K);
    LF_CHECK_LAUNCH("pointwise_fwd");
    return 0;
}
long lf_pointwise_scratch_floats(int N, int h, int w_, int C, int K) {
    const long npix = (long)N * h * w_;
    return (long)lf_cdiv(npix, 256) * (K * C + K);
}
// gx (N,h,w,C) NHWC or NULL; gw (K,C), gb (K) or NULL; scratch >= lf_pointwise_scratch_floats() floats
int lf_pointwise_bwd(const float* x, const float* gy, const float* w, float* gx, float* gw, float* gb, int N, int h, int w_,
                     int C, int K, float* scratch, void* stream) {
    LF_REQUIRE(x && gy && w && C % 4 == 0 && C <= PW_MAXC && K >= 1 && K <= PW_MAXK, "lf_pointwise_bwd: bad arguments (C=%d K=%d)", C, K);
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)N * h * w_;
    if (gx) {
        hipLaunchKernelGGL(pointwise_bwd_data_kernel, dim3(grid_for(npix, 4096)), dim3(256), 0, st, gy, w, gx, npix, (long)h * w_, C, K);
        LF_CHECK_LAUNCH("pointwise_bwd_data");
    }
    if (gw || gb) {       // weight and bias gradients are independent requests (a frozen weight with a trainable bias)
        LF_REQUIRE(scratch, "lf_pointwise_bwd: scratch missing");
        const int rows = lf_cdiv(npix, 256);
        float* brows = scratch + (long)rows * K * C;
        hipLaunchKernelGGL(pointwise_wgrad_kernel, dim3(rows), dim3(256), 0, st, x, gy, scratch, brows, npix, (long)h * w_, C, K, 256);
        LF_CHECK_LAUNCH("pointwise_wgrad");
        if (gw) LF_TRY(lf_rows_reduce_launch(scratch, rows, K * C, gw, 0, st));
        if (gb) LF_TRY(lf_rows_reduce_launch(brows, rows, K, gb, 0, st));
    }
    return 0;
}
}  // extern "C"
