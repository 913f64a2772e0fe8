// Fitting head and losses for gfx950: fused activation + row mask + weighted-least-squares
// normal equations (streaming fp64 moment accumulation), in-register solve, analytic
// backward, and the area / back-projection / cross-entropy losses.
//
// Reference behaviour (no code shared): BEV/Networks/LSQ_layer.py:103-167,310-325,
// BP/Networks/LSQ_layer.py:85-154, BP/Networks/gels.py, BEV/Loss_crit.py:61-134,
// BP/Loss_crit.py:166-218.   Math: SURVEY.md 2.2.
//
// Design (DESIGN.md "WLS layer"): the normal matrix of a polynomial fit is Hankel, so the
// whole (N,P,d+1) design-matrix / bmm chain of the reference collapses to 3d+2 moments
//   m_j = sum_i s_i y_i^j (j = 0..2d),  q_j = sum_i s_i x_i y_i^j (j = 0..d),  s_i = act(o_i)^2
// per (image, lane).  One pass over the logits (4 B/pixel) + the shared L2-resident grid;
// masked rows are never read.  Accumulation is fp64 (the reference's fp32 bmm over 131k
// pixels is the source of its 5e-5..1e-4 noise floor); HBM-bound.
#include "lf_common.h"

#include <stdarg.h>

thread_local char lf_err_buf[512] = "";
int lf_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(lf_err_buf, sizeof(lf_err_buf), fmt, ap);
    va_end(ap);
    return -1;
}
extern "C" const char* lf_last_error(void) { return lf_err_buf; }
extern "C" int lf_abi_version(void) { return LF_ABI_VERSION; }

namespace {

constexpr int WLS_THREADS = 256;
constexpr int WLS_CHUNKS = 16;   // row-chunks per (image, lane): N*K*16 workgroups >> 256 CUs at N*K = 64

__device__ __forceinline__ float act_fwd(float o, int kind) {
    switch (kind) {
        case LF_ACT_SQUARE: return o * o;
        case LF_ACT_ABS: return fabsf(o);
        case LF_ACT_RELU: return fmaxf(o, 0.f);
        case LF_ACT_SIGMOID: return 1.f / (1.f + expf(-o));
        case LF_ACT_SOFTPLUS: return o > 20.f ? o : log1pf(expf(o));
        default: return o;
    }
}
__device__ __forceinline__ float act_bwd(float o, int kind) {
    switch (kind) {
        case LF_ACT_SQUARE: return 2.f * o;
        case LF_ACT_ABS: return o > 0.f ? 1.f : (o < 0.f ? -1.f : 0.f);
        case LF_ACT_RELU: return o > 0.f ? 1.f : 0.f;
        case LF_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-o)); return s * (1.f - s); }
        case LF_ACT_SOFTPLUS: return o > 20.f ? 1.f : 1.f / (1.f + expf(-o));
        default: return 1.f;
    }
}

template <int ORDER>
struct Moments {
    static constexpr int NM = 2 * ORDER + 1, NQ = ORDER + 1, N = NM + NQ;
    double v[N];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = 0.0;
    }
    // one pixel: weight w (after activation), grid x', grid y', y = y_offset - y' in fp32 as the reference does
    __device__ __forceinline__ void add(float w, float gx, float gy, float y_off) {
        const double s = (double)w * (double)w;
        const double y = (double)(y_off - gy);
        const double x = (double)gx;
        double t = s;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            v[j] += t;
            if (j < NQ) v[NM + j] = fma(t, x, v[NM + j]);
            t *= y;
        }
    }
};

// Pass 1: per (chunk, image*lane) partial moments; optionally writes the masked weight map.
template <int ORDER, int VEC>
__global__ __launch_bounds__(WLS_THREADS) void wls_moments_kernel(
    const float* __restrict__ logits, const float* __restrict__ grid, long grid_bs, int K, int H, int W,
    int zero_rows, float y_off, int act_kind, float* __restrict__ masked, double* __restrict__ partials) {
    using M = Moments<ORDER>;
    const int nk = blockIdx.y, chunk = blockIdx.x;
    const long P = (long)H * W;
    const float* o = logits + (long)nk * P;
    const float* g = grid + (long)(nk / K) * grid_bs;
    float* mo = masked ? masked + (long)nk * P : nullptr;
    const long first = (long)zero_rows * W;             // first unmasked pixel
    const long units = (P - first) / VEC;
    const long u0 = units * chunk / WLS_CHUNKS, u1 = units * (chunk + 1) / WLS_CHUNKS;
    M acc;
    acc.zero();
    for (long u = u0 + threadIdx.x; u < u1; u += WLS_THREADS) {
        const long p = first + u * VEC;
        if constexpr (VEC == 4) {
            const float4 ov = *reinterpret_cast<const float4*>(o + p);
            const float4 g0 = *reinterpret_cast<const float4*>(g + 2 * p);
            const float4 g1 = *reinterpret_cast<const float4*>(g + 2 * p + 4);
            float4 w;
            w.x = act_fwd(ov.x, act_kind); w.y = act_fwd(ov.y, act_kind);
            w.z = act_fwd(ov.z, act_kind); w.w = act_fwd(ov.w, act_kind);
            acc.add(w.x, g0.x, g0.y, y_off);
            acc.add(w.y, g0.z, g0.w, y_off);
            acc.add(w.z, g1.x, g1.y, y_off);
            acc.add(w.w, g1.z, g1.w, y_off);
            if (mo) *reinterpret_cast<float4*>(mo + p) = w;
        } else {
            const float w = act_fwd(o[p], act_kind);
            acc.add(w, g[2 * p], g[2 * p + 1], y_off);
            if (mo) mo[p] = w;
        }
    }
    if (mo) {   // masked rows are zeros (index_fill), written without reading the logits
        const long zu = first / VEC, z0 = zu * chunk / WLS_CHUNKS, z1 = zu * (chunk + 1) / WLS_CHUNKS;
        for (long u = z0 + threadIdx.x; u < z1; u += WLS_THREADS) {
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(mo + u * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            else mo[u] = 0.f;
        }
    }
    __shared__ double red[WLS_THREADS / LF_WAVE][M::N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < M::N; ++j) {
        const double s = lf_wave_sum(acc.v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < M::N) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WLS_THREADS / LF_WAVE; ++w) s += red[w][threadIdx.x];
        partials[((long)nk * WLS_CHUNKS + chunk) * M::N + threadIdx.x] = s;
    }
}

// In-register inverse of a DxD matrix (Gauss-Jordan, partial pivoting = what LAPACK getrf/getri
// amount to for torch.inverse).  Returns 0 ok, 1 singular (zero / non-finite pivot).
template <int D>
__device__ int invert_lu(double (&A)[D][D], double (&Ai)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Ai[i][j] = (i == j) ? 1.0 : 0.0;
    int bad = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            const double a = fabs(A[r][c]);
            if (a > best) { best = a; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < D; ++r) {   // swap rows without dynamic register indexing
            if (r == piv) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = A[c][j]; A[c][j] = A[r][j]; A[r][j] = t;
                    t = Ai[c][j]; Ai[c][j] = Ai[r][j]; Ai[r][j] = t;
                }
            }
        }
        const double p = A[c][c];
        if (!(fabs(p) > 0.0) || !isfinite(p)) bad = 1;
        const double ip = 1.0 / p;
#pragma unroll
        for (int j = 0; j < D; ++j) { A[c][j] *= ip; Ai[c][j] *= ip; }
#pragma unroll
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const double f = A[r][c];
#pragma unroll
            for (int j = 0; j < D; ++j) { A[r][j] = fma(-f, A[c][j], A[r][j]); Ai[r][j] = fma(-f, Ai[c][j], Ai[r][j]); }
        }
    }
    return bad;
}

// Cholesky-based inverse (the GELS path).  Returns 0 ok, 2 when A is not positive definite.
template <int D>
__device__ int invert_chol(double (&A)[D][D], double (&Ai)[D][D]) {
    double L[D][D];
    int bad = 0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) L[i][j] = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0) || !isfinite(d)) bad = 2;
        const double ljj = sqrt(d);
        L[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / ljj;
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {   // solve L L^T x = e_c
        double y[D], x[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
#pragma unroll
        for (int i = D - 1; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int k = i + 1; k < D; ++k) s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) Ai[i][c] = x[i];
    }
    return bad;
}

// Pass 2: one thread per (image, lane): deterministic sum of the chunk partials, build the
// Hankel normal matrix, invert, beta = Z^-1 X.
template <int ORDER>
__global__ void wls_solve_kernel(const double* __restrict__ partials, int NK, double reg, int solver,
                                 double* __restrict__ beta, double* __restrict__ zinv, int32_t* __restrict__ status) {
    using M = Moments<ORDER>;
    constexpr int D = ORDER + 1;
    const int nk = blockIdx.x * blockDim.x + threadIdx.x;
    if (nk >= NK) return;
    double mom[M::N];
#pragma unroll
    for (int j = 0; j < M::N; ++j) mom[j] = 0.0;
    for (int c = 0; c < WLS_CHUNKS; ++c)
#pragma unroll
        for (int j = 0; j < M::N; ++j) mom[j] += partials[((long)nk * WLS_CHUNKS + c) * M::N + j];
    double Z[D][D], Zi[D][D], X[D];
    const double r = reg;     // both solvers (BEV/Networks/LSQ_layer.py:120-126); the GELS flavour's caller passes 0 (gels.py has none)
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) Z[i][j] = mom[(ORDER - i) + (ORDER - j)] + (i == j ? r : 0.0);
        X[i] = mom[M::NM + (ORDER - i)];
    }
    const int st = (solver == LF_SOLVE_CHOLESKY) ? invert_chol<D>(Z, Zi) : invert_lu<D>(Z, Zi);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double b = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            b = fma(Zi[i][j], X[j], b);
            zinv[((long)nk * D + i) * D + j] = Zi[i][j];
        }
        beta[(long)nk * D + i] = b;
    }
    status[nk] = st;
}

// Backward: d/d logits of sum_k <grad_beta_k, beta_k>.  Elementwise over the whole map.
template <int ORDER, int VEC>
__global__ __launch_bounds__(WLS_THREADS) void wls_bwd_kernel(
    const float* __restrict__ logits, const float* __restrict__ grid, long grid_bs, int K, int H, int W,
    int zero_rows, float y_off, int act_kind, const double* __restrict__ beta, const double* __restrict__ zinv,
    const double* __restrict__ gbeta, float* __restrict__ gout) {
    constexpr int D = ORDER + 1;
    const int nk = blockIdx.y;
    const long P = (long)H * W;
    const float* o = logits + (long)nk * P;
    const float* g = grid + (long)(nk / K) * grid_bs;
    float* go = gout + (long)nk * P;
    double b[D], v[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        b[i] = beta[(long)nk * D + i];
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s = fma(zinv[((long)nk * D + i) * D + j], gbeta[(long)nk * D + j], s);
        v[i] = s;   // Z^-1 is symmetric, so Z^-T g = Z^-1 g
    }
    const long first = (long)zero_rows * W;
    auto one = [&](float ov, float gx, float gy) -> float {
        const double y = (double)(y_off - gy);
        double yv = v[0], yb = b[0];
#pragma unroll
        for (int i = 1; i < D; ++i) { yv = fma(yv, y, v[i]); yb = fma(yb, y, b[i]); }   // Horner, highest power first
        const double w = (double)act_fwd(ov, act_kind);
        return (float)(2.0 * w * yv * ((double)gx - yb) * (double)act_bwd(ov, act_kind));
    };
    const long units = P / VEC;
    for (long u = (long)blockIdx.x * WLS_THREADS + threadIdx.x; u < units; u += (long)gridDim.x * WLS_THREADS) {
        const long p = u * VEC;
        if constexpr (VEC == 4) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p >= first) {
                const float4 ov = *reinterpret_cast<const float4*>(o + p);
                const float4 g0 = *reinterpret_cast<const float4*>(g + 2 * p);
                const float4 g1 = *reinterpret_cast<const float4*>(g + 2 * p + 4);
                r.x = one(ov.x, g0.x, g0.y); r.y = one(ov.y, g0.z, g0.w);
                r.z = one(ov.z, g1.x, g1.y); r.w = one(ov.w, g1.z, g1.w);
            }
            *reinterpret_cast<float4*>(go + p) = r;
        } else {
            go[p] = (p >= first) ? one(o[p], g[2 * p], g[2 * p + 1]) : 0.f;
        }
    }
}

template <int ORDER>
int wls_fwd_launch(const float* logits, const float* grid, long gbs, int N, int K, int H, int W, int zr,
                   double reg, double y_off, int act, int solver, double* beta, double* zinv, float* masked,
                   double* partials, int32_t* status, hipStream_t st) {
    const bool vec = (W % 4 == 0);
    dim3 g1(WLS_CHUNKS, N * K);
    if (vec)
        hipLaunchKernelGGL((wls_moments_kernel<ORDER, 4>), g1, dim3(WLS_THREADS), 0, st, logits, grid, gbs, K, H, W,
                           zr, (float)y_off, act, masked, partials);
    else
        hipLaunchKernelGGL((wls_moments_kernel<ORDER, 1>), g1, dim3(WLS_THREADS), 0, st, logits, grid, gbs, K, H, W,
                           zr, (float)y_off, act, masked, partials);
    LF_CHECK_LAUNCH("wls_moments");
    hipLaunchKernelGGL((wls_solve_kernel<ORDER>), dim3(lf_cdiv(N * K, 64)), dim3(64), 0, st, partials, N * K, reg,
                       solver, beta, zinv, status);
    LF_CHECK_LAUNCH("wls_solve");
    return 0;
}

template <int ORDER>
int wls_bwd_launch(const float* logits, const float* grid, long gbs, int N, int K, int H, int W, int zr,
                   double y_off, int act, const double* beta, const double* zinv, const double* gbeta, float* gout,
                   hipStream_t st) {
    const bool vec = (W % 4 == 0);
    const long units = (long)H * W / (vec ? 4 : 1);
    int gx = lf_cdiv(units, WLS_THREADS);
    if (gx > 64) gx = 64;
    dim3 g1(gx, N * K);
    if (vec)
        hipLaunchKernelGGL((wls_bwd_kernel<ORDER, 4>), g1, dim3(WLS_THREADS), 0, st, logits, grid, gbs, K, H, W, zr,
                           (float)y_off, act, beta, zinv, gbeta, gout);
    else
        hipLaunchKernelGGL((wls_bwd_kernel<ORDER, 1>), g1, dim3(WLS_THREADS), 0, st, logits, grid, gbs, K, H, W, zr,
                           (float)y_off, act, beta, zinv, gbeta, gout);
    LF_CHECK_LAUNCH("wls_bwd");
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// GELS: least squares of an explicit design matrix through the normal equations + Cholesky
// (BP/Networks/gels.py:9-25).  A (N,P,D) fp32, b (N,P) fp32, D <= 4.
// ---------------------------------------------------------------------------------------
namespace {

template <int D>
__global__ __launch_bounds__(WLS_THREADS) void gels_moments_kernel(const float* __restrict__ A, const float* __restrict__ b,
                                                                  long P, double* __restrict__ partials) {
    constexpr int NS = D * (D + 1) / 2 + D;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const long p0 = P * chunk / WLS_CHUNKS, p1 = P * (chunk + 1) / WLS_CHUNKS;
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
    for (long p = p0 + threadIdx.x; p < p1; p += WLS_THREADS) {
        double a[D];
#pragma unroll
        for (int j = 0; j < D; ++j) a[j] = (double)A[((long)n * P + p) * D + j];
        const double bv = (double)b[(long)n * P + p];
        int k = 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) { acc[k] = fma(a[i], a[j], acc[k]); ++k; }
#pragma unroll
        for (int i = 0; i < D; ++i) acc[D * (D + 1) / 2 + i] = fma(a[i], bv, acc[D * (D + 1) / 2 + i]);
    }
    __shared__ double red[WLS_THREADS / LF_WAVE][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const double v = lf_wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < WLS_THREADS / LF_WAVE; ++w) v += red[w][threadIdx.x];
        partials[((long)n * WLS_CHUNKS + chunk) * NS + threadIdx.x] = v;
    }
}

template <int D>
__global__ void gels_solve_kernel(const double* __restrict__ partials, int N, float* __restrict__ x,
                                  double* __restrict__ zinv, int32_t* __restrict__ status) {
    constexpr int NS = D * (D + 1) / 2 + D;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double m[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) m[j] = 0.0;
    for (int c = 0; c < WLS_CHUNKS; ++c)
#pragma unroll
        for (int j = 0; j < NS; ++j) m[j] += partials[((long)n * WLS_CHUNKS + c) * NS + j];
    double Z[D][D], Zi[D][D], X[D];
    int k = 0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) { Z[i][j] = m[k]; Z[j][i] = m[k]; ++k; }
#pragma unroll
    for (int i = 0; i < D; ++i) X[i] = m[D * (D + 1) / 2 + i];
    const int st = invert_chol<D>(Z, Zi);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) { v = fma(Zi[i][j], X[j], v); zinv[((long)n * D + i) * D + j] = Zi[i][j]; }
        x[(long)n * D + i] = (float)v;
    }
    status[n] = st;
}

template <int D>
__global__ __launch_bounds__(WLS_THREADS) void gels_bwd_kernel(const float* __restrict__ A, const float* __restrict__ b,
                                                              const float* __restrict__ x, const double* __restrict__ zinv,
                                                              const float* __restrict__ gout, long P,
                                                              float* __restrict__ gA, float* __restrict__ gb) {
    const int n = blockIdx.y;
    double xs[D], z[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        xs[i] = (double)x[(long)n * D + i];
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) v = fma(zinv[((long)n * D + i) * D + j], (double)gout[(long)n * D + j], v);
        z[i] = v;
    }
    for (long p = (long)blockIdx.x * WLS_THREADS + threadIdx.x; p < P; p += (long)gridDim.x * WLS_THREADS) {
        double a[D], ax = 0.0, az = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) { a[j] = (double)A[((long)n * P + p) * D + j]; ax = fma(a[j], xs[j], ax); az = fma(a[j], z[j], az); }
        const double bv = (double)b[(long)n * P + p];
#pragma unroll
        for (int j = 0; j < D; ++j) gA[((long)n * P + p) * D + j] = (float)(-(ax * z[j] + az * xs[j]) + bv * z[j]);
        gb[(long)n * P + p] = (float)az;
    }
}

}  // namespace

extern "C" size_t lf_gels_workspace_bytes(int N, int D) { return (size_t)N * WLS_CHUNKS * (D * (D + 1) / 2 + D) * sizeof(double); }

extern "C" int lf_gels_fwd(const float* A, const float* b, int N, long P, int D, float* x, double* zinv, void* partials,
                           int32_t* status, void* stream) {
    LF_REQUIRE(A && b && x && zinv && partials && status, "lf_gels_fwd: null pointer");
    LF_REQUIRE(D >= 1 && D <= 4 && N > 0 && P > 0, "lf_gels_fwd: bad shape N=%d P=%ld D=%d", N, P, D);
    hipStream_t st = (hipStream_t)stream;
    dim3 g1(WLS_CHUNKS, N);
    double* pp = (double*)partials;
#define LF_GELS(DD)                                                                                              \
    hipLaunchKernelGGL(gels_moments_kernel<DD>, g1, dim3(WLS_THREADS), 0, st, A, b, P, pp);                      \
    hipLaunchKernelGGL(gels_solve_kernel<DD>, dim3(lf_cdiv(N, 64)), dim3(64), 0, st, pp, N, x, zinv, status)
    switch (D) {
        case 1: LF_GELS(1); break;
        case 2: LF_GELS(2); break;
        case 3: LF_GELS(3); break;
        default: LF_GELS(4); break;
    }
#undef LF_GELS
    LF_CHECK_LAUNCH("gels_fwd");
    return 0;
}

extern "C" int lf_gels_bwd(const float* A, const float* b, const float* x, const double* zinv, const float* grad_out, int N,
                           long P, int D, float* grad_A, float* grad_b, void* stream) {
    LF_REQUIRE(A && b && x && zinv && grad_out && grad_A && grad_b, "lf_gels_bwd: null pointer");
    LF_REQUIRE(D >= 1 && D <= 4 && N > 0 && P > 0, "lf_gels_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    int gx = lf_cdiv(P, WLS_THREADS);
    if (gx > 256) gx = 256;
    dim3 g1(gx, N);
    switch (D) {
        case 1: hipLaunchKernelGGL(gels_bwd_kernel<1>, g1, dim3(WLS_THREADS), 0, st, A, b, x, zinv, grad_out, P, grad_A, grad_b); break;
        case 2: hipLaunchKernelGGL(gels_bwd_kernel<2>, g1, dim3(WLS_THREADS), 0, st, A, b, x, zinv, grad_out, P, grad_A, grad_b); break;
        case 3: hipLaunchKernelGGL(gels_bwd_kernel<3>, g1, dim3(WLS_THREADS), 0, st, A, b, x, zinv, grad_out, P, grad_A, grad_b); break;
        default: hipLaunchKernelGGL(gels_bwd_kernel<4>, g1, dim3(WLS_THREADS), 0, st, A, b, x, zinv, grad_out, P, grad_A, grad_b); break;
    }
    LF_CHECK_LAUNCH("gels_bwd");
    return 0;
}

extern "C" size_t lf_wls_workspace_bytes(int N, int K, int order) {
    return (size_t)N * K * WLS_CHUNKS * (3 * order + 2) * sizeof(double);
}

extern "C" int lf_wls_fwd(const float* logits, const float* grid_xy, long grid_batch_stride, int N, int K, int H,
                          int W, int zero_rows, int order, double reg, double y_offset, int act_kind, int solver,
                          double* beta, double* zinv, float* masked, void* partials, int32_t* status, void* stream) {
    LF_REQUIRE(logits && grid_xy && beta && zinv && partials && status, "lf_wls_fwd: null pointer");
    LF_REQUIRE(N > 0 && K > 0 && H > 0 && W > 0, "lf_wls_fwd: bad shape %d %d %d %d", N, K, H, W);
    LF_REQUIRE(zero_rows >= 0 && zero_rows < H, "lf_wls_fwd: zero_rows %d out of [0,%d)", zero_rows, H);
    LF_REQUIRE(order >= 0 && order <= 3, "lf_wls_fwd: order %d not in 0..3", order);
    LF_REQUIRE(act_kind >= 0 && act_kind <= LF_ACT_NONE, "lf_wls_fwd: bad activation %d", act_kind);
    hipStream_t st = (hipStream_t)stream;
    double* p = (double*)partials;
    switch (order) {
        case 0: return wls_fwd_launch<0>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, reg, y_offset, act_kind, solver, beta, zinv, masked, p, status, st);
        case 1: return wls_fwd_launch<1>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, reg, y_offset, act_kind, solver, beta, zinv, masked, p, status, st);
        case 2: return wls_fwd_launch<2>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, reg, y_offset, act_kind, solver, beta, zinv, masked, p, status, st);
        default: return wls_fwd_launch<3>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, reg, y_offset, act_kind, solver, beta, zinv, masked, p, status, st);
    }
}

extern "C" int lf_wls_bwd(const float* logits, const float* grid_xy, long grid_batch_stride, int N, int K, int H,
                          int W, int zero_rows, int order, double y_offset, int act_kind, const double* beta,
                          const double* zinv, const double* grad_beta, float* grad_logits, void* stream) {
    LF_REQUIRE(logits && grid_xy && beta && zinv && grad_beta && grad_logits, "lf_wls_bwd: null pointer");
    LF_REQUIRE(N > 0 && K > 0 && H > 0 && W > 0, "lf_wls_bwd: bad shape");
    LF_REQUIRE(order >= 0 && order <= 3, "lf_wls_bwd: order %d not in 0..3", order);
    hipStream_t st = (hipStream_t)stream;
    switch (order) {
        case 0: return wls_bwd_launch<0>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, y_offset, act_kind, beta, zinv, grad_beta, grad_logits, st);
        case 1: return wls_bwd_launch<1>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, y_offset, act_kind, beta, zinv, grad_beta, grad_logits, st);
        case 2: return wls_bwd_launch<2>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, y_offset, act_kind, beta, zinv, grad_beta, grad_logits, st);
        default: return wls_bwd_launch<3>(logits, grid_xy, grid_batch_stride, N, K, H, W, zero_rows, y_offset, act_kind, beta, zinv, grad_beta, grad_logits, st);
    }
}

// ---------------------------------------------------------------------------------------
// Area loss (single workgroup; N is a batch size)
// ---------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void area_loss_kernel(const T* __restrict__ beta, long bstride, const T* __restrict__ gt,
                                                       int N, int order, int wf, T* __restrict__ loss, T* __restrict__ grad) {
    const int D = order + 1;
    const double t = 0.7;
    const double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t, t6 = t5 * t;
    const double t15 = pow(t, 1.5), t25 = pow(t, 2.5), t35 = pow(t, 3.5), t45 = pow(t, 4.5), t55 = pow(t, 5.5);
    __shared__ double sL[4], sC[4];
    __shared__ double tot[2];
    double Lsum = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        double d[3] = {0, 0, 0};
        bool keep = true;
        for (int j = 0; j < D; ++j) {
            const double gj = (double)gt[(long)i * D + j];
            d[j] = (double)beta[(long)i * bstride + j] - gj;
            keep = keep && (gj != 0.0);
        }
        const double a = d[0], b = d[1], c = d[2];
        double L;
        if (order == 2) {
            if (wf == LF_WF_NONE)
                L = a * a * t5 / 5 + 2 * a * b * t4 / 4 + (b * b + c * 2 * a) * t3 / 3 + 2 * b * c * t2 / 2 + c * c * t;
            else if (wf == LF_WF_LINEAR)
                L = c * c * t - t5 * ((2 * a * b) / 5 - a * a / 5) + t2 * (b * c - c * c / 2) - (a * a * t6) / 6 -
                    t4 * (b * b / 4 - (a * b) / 2 + (a * c) / 2) + t3 * (b * b / 3 - (2 * c * b) / 3 + (2 * a * c) / 3);
            else
                L = t3 * (b * b / 3 + 2.0 / 3 * a * c) - t35 * (2.0 / 7 * b * b + 4.0 / 7 * a * c) + c * c * t +
                    0.2 * a * a * t5 - 2.0 / 11 * a * a * t55 - 2.0 / 3 * c * c * t15 + 0.5 * a * b * t4 -
                    4.0 / 9 * a * b * t45 + b * c * t2 - 0.8 * b * c * t25;
        } else {
            L = b * b * t + a * b * t2 + (a * a * t3) / 3;
        }
        if (keep) { Lsum += L; cnt += 1.0; }
    }
    Lsum = lf_wave_sum(Lsum);
    cnt = lf_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { sL[threadIdx.x >> 6] = Lsum; sC[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tot[0] = sL[0] + sL[1] + sL[2] + sL[3];
        tot[1] = sC[0] + sC[1] + sC[2] + sC[3];
        loss[0] = (T)(tot[1] > 0 ? tot[0] / tot[1] : 0.0);
    }
    __syncthreads();
    const double inv = tot[1] > 0 ? 1.0 / tot[1] : 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        double d[3] = {0, 0, 0};
        bool keep = true;
        for (int j = 0; j < D; ++j) {
            const double gj = (double)gt[(long)i * D + j];
            d[j] = (double)beta[(long)i * bstride + j] - gj;
            keep = keep && (gj != 0.0);
        }
        const double a = d[0], b = d[1], c = d[2];
        double g[3] = {0, 0, 0};
        if (order == 2) {
            if (wf == LF_WF_NONE) {
                g[0] = 2 * a * t5 / 5 + b * t4 / 2 + 2 * c * t3 / 3;
                g[1] = a * t4 / 2 + 2 * b * t3 / 3 + c * t2;
                g[2] = 2 * a * t3 / 3 + b * t2 + 2 * c * t;
            } else if (wf == LF_WF_LINEAR) {
                g[0] = -t5 * (2 * b / 5 - 2 * a / 5) - a * t6 / 3 - t4 * (-b / 2 + c / 2) + t3 * (2 * c / 3);
                g[1] = -t5 * (2 * a / 5) + t2 * c - t4 * (b / 2 - a / 2) + t3 * (2 * b / 3 - 2 * c / 3);
                g[2] = 2 * c * t + t2 * (b - c) - t4 * (a / 2) + t3 * (-2 * b / 3 + 2 * a / 3);
            } else {
                g[0] = t3 * (2.0 / 3 * c) - t35 * (4.0 / 7 * c) + 0.4 * a * t5 - 4.0 / 11 * a * t55 + 0.5 * b * t4 - 4.0 / 9 * b * t45;
                g[1] = t3 * (2.0 / 3 * b) - t35 * (4.0 / 7 * b) + 0.5 * a * t4 - 4.0 / 9 * a * t45 + c * t2 - 0.8 * c * t25;
                g[2] = t3 * (2.0 / 3 * a) - t35 * (4.0 / 7 * a) + 2 * c * t - 4.0 / 3 * c * t15 + b * t2 - 0.8 * b * t25;
            }
        } else {
            g[0] = b * t2 + 2 * a * t3 / 3;
            g[1] = 2 * b * t + a * t2;
        }
        for (int j = 0; j < D; ++j) grad[(long)i * D + j] = (T)(keep ? g[j] * inv : 0.0);
    }
}

// Back-projection loss: thread per image row of S sample heights; two phases in one workgroup.
__global__ __launch_bounds__(256) void backproj_kernel(const double* __restrict__ beta, long bstride,
                                                      const double* __restrict__ x_gt, const double* __restrict__ valid,
                                                      const double* __restrict__ Y, const double* __restrict__ yp,
                                                      double m00, double m01, double m02, double m20, double m21, double m22,
                                                      int N, int S, int order, double* __restrict__ loss,
                                                      double* __restrict__ xcv, double* __restrict__ grad) {
    const int D = order + 1;
    __shared__ double sE[4], sV[4];
    __shared__ double tot[2];
    double e2 = 0.0, nv = 0.0;
    for (long idx = threadIdx.x; idx < (long)N * S; idx += 256) {
        const int n = (int)(idx / S), j = (int)(idx % S);
        double xp = 0.0;
        for (int i = 0; i < D; ++i) xp = fma(Y[(long)j * D + i], beta[(long)n * bstride + i], xp);
        const double t0 = m00 * xp + m01 * yp[j] + m02;
        const double t2 = m20 * xp + m21 * yp[j] + m22;
        const double xc = t0 / t2;
        const double v = valid[idx];
        const double err = (x_gt[idx] - xc) * v;
        xcv[idx] = xc * v;
        e2 = fma(err, err, e2);
        nv += v;
    }
    e2 = lf_wave_sum(e2);
    nv = lf_wave_sum(nv);
    if ((threadIdx.x & 63) == 0) { sE[threadIdx.x >> 6] = e2; sV[threadIdx.x >> 6] = nv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tot[0] = sE[0] + sE[1] + sE[2] + sE[3];
        tot[1] = sV[0] + sV[1] + sV[2] + sV[3];
        loss[0] = tot[1] != 0.0 ? tot[0] / tot[1] : 0.0;
    }
    __syncthreads();
    const double inv = tot[1] != 0.0 ? 1.0 / tot[1] : 0.0;
    // gradient: FOUR lanes per image, each taking every fourth sample height, merged in a fixed order by two butterfly steps
    // (round 6: one lane per image walked all 56 heights -- 112 dependent fp64 divisions on ONE wave, 43 us per launch and four
    // launches per step at BASELINE config 3)
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int n = n0 + (int)(threadIdx.x >> 2), q = (int)(threadIdx.x & 3);
        const bool live = n < N;
        const int nn = live ? n : 0;
        double g[4] = {0, 0, 0, 0};
        for (int j = q; j < S && live; j += 4) {
            double xp = 0.0;
            for (int i = 0; i < D; ++i) xp = fma(Y[(long)j * D + i], beta[(long)nn * bstride + i], xp);
            const double t0 = m00 * xp + m01 * yp[j] + m02;
            const double t2 = m20 * xp + m21 * yp[j] + m22;
            const double v = valid[(long)nn * S + j];
            const double err = (x_gt[(long)nn * S + j] - t0 / t2) * v;
            const double dxc = (m00 * t2 - m20 * t0) / (t2 * t2);
            const double gx = -2.0 * err * v * inv * dxc;
            for (int i = 0; i < D; ++i) g[i] = fma(gx, Y[(long)j * D + i], g[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g[i] += __shfl_xor(g[i], 1, 64);
            g[i] += __shfl_xor(g[i], 2, 64);
        }
        if (live && q == 0)
            for (int i = 0; i < D; ++i) grad[(long)n * D + i] = g[i];
    }
}

// Cross entropy: one thread per pixel, NCHW logits (channel planes are contiguous in w => coalesced).
constexpr int CE_MAXC = 8;
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ tgt,
                                                    const float* __restrict__ wts, int C, long HW, long total,
                                                    double* __restrict__ acc) {
    double num = 0.0, den = 0.0, nbad = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i % HW;
        const float* zp = z + n * C * HW + p;
        float v[CE_MAXC], mx = -INFINITY;
        for (int c = 0; c < C; ++c) { v[c] = zp[(long)c * HW]; mx = fmaxf(mx, v[c]); }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(v[c] - mx);
        const int64_t t64 = tgt[i];
        const bool okt = t64 >= 0 && t64 < C;               // a label outside [0, C) is counted and ignored (weight 0)
        const int t = okt ? (int)t64 : 0;
        float zt = 0.f;
        for (int c = 0; c < C; ++c) zt = (c == t) ? v[c] : zt;
        const float w = okt ? wts[t] : 0.f;
        num += (double)(w * (mx + logf(se) - zt));
        den += (double)w;
        nbad += okt ? 0.0 : 1.0;
    }
    num = lf_wave_sum(num);
    den = lf_wave_sum(den);
    nbad = lf_wave_sum(nbad);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(acc, num); atomicAdd(acc + 1, den);
        if (nbad != 0.0) atomicAdd(acc + 2, nbad);
    }
}
__global__ void ce_finish_kernel(const double* acc, float* loss) { loss[0] = (float)(acc[0] / acc[1]); }
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ tgt,
                                                    const float* __restrict__ wts, int C, long HW, long total,
                                                    const double* __restrict__ acc, const float* __restrict__ up,
                                                    float* __restrict__ gz) {
    const float scale = up[0] / (float)acc[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i % HW;
        const float* zp = z + n * C * HW + p;
        float* gp = gz + n * C * HW + p;
        float v[CE_MAXC], mx = -INFINITY;
        for (int c = 0; c < C; ++c) { v[c] = zp[(long)c * HW]; mx = fmaxf(mx, v[c]); }
        float se = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - mx); se += v[c]; }
        const int64_t t64 = tgt[i];
        const bool okt = t64 >= 0 && t64 < C;
        const int t = okt ? (int)t64 : 0;
        const float w = okt ? wts[t] * scale : 0.f, ise = 1.f / se;
        for (int c = 0; c < C; ++c) gp[(long)c * HW] = w * (v[c] * ise - (c == t ? 1.f : 0.f));
    }
}

}  // namespace

extern "C" int lf_area_loss(const void* beta, long beta_stride, const void* gt, int N, int order, int weight_funct,
                            int dtype, void* loss, void* grad, void* stream) {
    LF_REQUIRE(beta && gt && loss && grad, "lf_area_loss: null pointer");
    LF_REQUIRE(order == 1 || order == 2, "lf_area_loss: order %d not implemented (reference: Loss_crit.py:125-128)", order);
    LF_REQUIRE(weight_funct >= 0 && weight_funct <= 2, "lf_area_loss: bad weight function %d", weight_funct);
    LF_REQUIRE(N > 0, "lf_area_loss: N must be positive");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LF_F64)
        hipLaunchKernelGGL(area_loss_kernel<double>, dim3(1), dim3(256), 0, st, (const double*)beta, beta_stride,
                           (const double*)gt, N, order, weight_funct, (double*)loss, (double*)grad);
    else
        hipLaunchKernelGGL(area_loss_kernel<float>, dim3(1), dim3(256), 0, st, (const float*)beta, beta_stride,
                           (const float*)gt, N, order, weight_funct, (float*)loss, (float*)grad);
    LF_CHECK_LAUNCH("area_loss");
    return 0;
}

namespace {
// MSE_Loss: mean over all n elements of (p - q)^2 and its gradient 2 (p - q) / n; one block, fp64 accumulation
template <typename T>
__global__ __launch_bounds__(256) void mse_loss_kernel(const T* __restrict__ p, const T* __restrict__ q, long n,
                                                      T* __restrict__ loss, T* __restrict__ grad) {
    __shared__ double sw[4];
    double acc = 0.0;
    const double inv = 1.0 / (double)n;
    for (long i = threadIdx.x; i < n; i += 256) {
        const double d = (double)p[i] - (double)q[i];
        acc += d * d;
        grad[i] = (T)(2.0 * d * inv);
    }
    acc = lf_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (T)(((sw[0] + sw[1]) + (sw[2] + sw[3])) * inv);
}
}  // namespace

extern "C" int lf_mse_loss(const void* params, const void* gt, long n, int dtype, void* loss, void* grad, void* stream) {
    LF_REQUIRE(params && gt && loss && grad && n > 0, "lf_mse_loss: bad arguments");
    if (dtype == LF_F64)
        hipLaunchKernelGGL(mse_loss_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)params, (const double*)gt, n,
                           (double*)loss, (double*)grad);
    else
        hipLaunchKernelGGL(mse_loss_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)params, (const float*)gt, n,
                           (float*)loss, (float*)grad);
    LF_CHECK_LAUNCH("mse_loss");
    return 0;
}

extern "C" int lf_backproj_loss(const double* beta, long beta_stride, const double* x_gt, const double* valid,
                                const double* Y, const double* y_prime, const double* minv_host, int N, int S,
                                int order, double* loss, double* x_cal_valid, double* grad, void* stream) {
    LF_REQUIRE(beta && x_gt && valid && Y && y_prime && minv_host && loss && x_cal_valid && grad,
               "lf_backproj_loss: null pointer");
    LF_REQUIRE(order >= 0 && order <= 3 && N > 0 && S > 0, "lf_backproj_loss: bad shape");
    const double* m = minv_host;
    hipLaunchKernelGGL(backproj_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, beta, beta_stride, x_gt, valid, Y,
                       y_prime, m[0], m[1], m[2], m[6], m[7], m[8], N, S, order, loss, x_cal_valid, grad);
    LF_CHECK_LAUNCH("backproj_loss");
    return 0;
}

extern "C" int lf_ce2d_fwd(const float* logits, const int64_t* target, const float* weights, int N, int C, int H, int W,
                           double* acc, float* loss, void* stream) {
    LF_REQUIRE(logits && target && weights && acc && loss, "lf_ce2d_fwd: null pointer");
    LF_REQUIRE(C >= 1 && C <= CE_MAXC, "lf_ce2d_fwd: C=%d not in 1..%d", C, CE_MAXC);
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W, total = (long)N * HW;
    if (hipMemsetAsync(acc, 0, 3 * sizeof(double), st) != hipSuccess) return lf_fail("lf_ce2d_fwd: memset failed");
    int grid = lf_cdiv(total, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid), dim3(256), 0, st, logits, target, weights, C, HW, total, acc);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(1), 0, st, acc, loss);
    LF_CHECK_LAUNCH("ce2d_fwd");
    return 0;
}

extern "C" int lf_ce2d_bwd(const float* logits, const int64_t* target, const float* weights, int N, int C, int H, int W,
                           const double* acc, const float* upstream, float* grad_logits, void* stream) {
    LF_REQUIRE(logits && target && weights && acc && upstream && grad_logits, "lf_ce2d_bwd: null pointer");
    LF_REQUIRE(C >= 1 && C <= CE_MAXC, "lf_ce2d_bwd: C=%d not in 1..%d", C, CE_MAXC);
    const long HW = (long)H * W, total = (long)N * HW;
    int grid = lf_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, weights, C, HW,
                       total, acc, upstream, grad_logits);
    LF_CHECK_LAUNCH("ce2d_bwd");
    return 0;
}

// ---- inference-side back-projection + lane post-processing (BP/test.py:60-88, Projections :128-186) -------
// One thread per (image, lane, sample height): evaluate the fitted polynomial at y_eval, map (x', y') back
// through M_inv, scale to the 1280-wide frame, then apply the three gates of test_model in its order:
// line-type flag == 0 -> fill; sample index below the horizon bound -> fill; outside [lo, hi] -> fill.
__global__ __launch_bounds__(256) void lane_decode_kernel(const double* __restrict__ beta, const double* __restrict__ y_eval,
                                                         const double* __restrict__ y_prime, double m0, double m1, double m2,
                                                         double m6, double m7, double m8, double scale,
                                                         const float* __restrict__ line_flag, const int* __restrict__ bound,
                                                         double lo, double hi, double fill, int N, int L, int S, int order,
                                                         double* __restrict__ x_out, int* __restrict__ x_int) {
    const long total = (long)N * L * S;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int s = (int)(u % S);
        const long nl = u / S;
        const int n = (int)(nl / L);
        const double* b = beta + nl * (order + 1);
        const double ye = y_eval[s];
        double xp = b[0];
        for (int k = 1; k <= order; ++k) xp = xp * ye + b[k];          // highest power first, as Projections.Y
        const double yp = y_prime[s];
        double x = (m0 * xp + m1 * yp + m2) / (m6 * xp + m7 * yp + m8) * scale;
        if (line_flag && line_flag[nl] == 0.f) x = fill;
        if (bound) {
            int bd = bound[n];                                          // python slice [:bd]
            if (bd < 0) bd = S + bd < 0 ? 0 : S + bd;
            if (s < bd) x = fill;
        }
        if (x > hi) x = fill;
        if (x < lo) x = fill;
        if (x_out) x_out[u] = x;
        if (x_int) x_int[u] = (int)rint(x);                             // np.round: half to even
    }
}

// beta (N, L, order+1) fp64 contiguous; line_flag (N, L) fp32 (already in lane order) or NULL; bound (N) int32 or NULL;
// lo > hi disables the range gate.  x_out (N, L, S) fp64 and/or x_int (N, L, S) int32.
extern "C" int lf_lane_decode(const double* beta, const double* y_eval, const double* y_prime, const double* minv_host,
                              double scale, const float* line_flag, const int* bound, double lo, double hi, double fill, int N,
                              int L, int S, int order, double* x_out, int* x_int, void* stream) {
    LF_REQUIRE(beta && y_eval && y_prime && minv_host && (x_out || x_int), "lf_lane_decode: null pointer");
    LF_REQUIRE(order >= 0 && order <= 3 && N > 0 && L > 0 && S > 0, "lf_lane_decode: bad shape");
    const double* m = minv_host;
    if (lo > hi) { lo = -1e300; hi = 1e300; }
    hipLaunchKernelGGL(lane_decode_kernel, dim3(lf_cdiv((long)N * L * S, 256)), dim3(256), 0, (hipStream_t)stream, beta, y_eval,
                       y_prime, m[0], m[1], m[2], m[6], m[7], m[8], scale, line_flag, bound, lo, hi, fill, N, L, S, order, x_out,
                       x_int);
    LF_CHECK_LAUNCH("lane_decode");
    return 0;
}

// ---- exact-area metric (BEV/Loss_crit.py:12-35 polynomial.trapezoidal) ------------------------------------
// One thread per curve pair; the sum runs in the reference's order and dtype so fp32 inputs round identically.
template <typename T>
__global__ __launch_bounds__(256) void trapezoid_kernel(const T* __restrict__ p, const T* __restrict__ q, int B, double a,
                                                       double b, int n, T* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const T pa = p[3 * i], pb = p[3 * i + 1], pc = p[3 * i + 2];
    const T qa = q[3 * i], qb = q[3 * i + 1], qc = q[3 * i + 2];
    const double h = (b - a) / n;
    auto ev = [](T c2, T c1, T c0, double x) -> T { return c2 * (T)(x * x) + c1 * (T)x + c0; };
    T s = (T)0;
    s += fabs(ev(pa, pb, pc, a) / (T)2 - ev(qa, qb, qc, a) / (T)2);
    for (int k = 1; k < n; ++k) s += fabs(ev(pa, pb, pc, a + k * h) - ev(qa, qb, qc, a + k * h));
    s += fabs(ev(pa, pb, pc, b) / (T)2 - ev(qa, qb, qc, b) / (T)2);
    out[i] = s * (T)h;
}

// p, q: (B, 3) coefficient rows [a, b, c] of a*x^2 + b*x + c; is_double selects fp64 / fp32 storage.
extern "C" int lf_trapezoid(const void* p, const void* q, int B, double a, double b, int n, int is_double, void* out,
                            void* stream) {
    LF_REQUIRE(p && q && out && B > 0 && n > 0, "lf_trapezoid: bad arguments");
    if (is_double)
        hipLaunchKernelGGL(trapezoid_kernel<double>, dim3(lf_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)p,
                           (const double*)q, B, a, b, n, (double*)out);
    else
        hipLaunchKernelGGL(trapezoid_kernel<float>, dim3(lf_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)p,
                           (const float*)q, B, a, b, n, (float*)out);
    LF_CHECK_LAUNCH("trapezoid");
    return 0;
}
