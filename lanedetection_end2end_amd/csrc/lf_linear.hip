// Cold-path arithmetic that round 3 still left to ATen / rocBLAS (Missing #6), as plain HIP kernels:
//
//  * lf_linear_fwd / lf_linear_bwd: the nn.Linear tails of the --clas heads -- fully_connected1 (32768 -> 128, + ReLU),
//    fully_connected_line1 (128 -> 4) / fully_connected_line1..4 (128 -> 3 each), fully_connected_horizon (2048 -> resize)
//    (BP/Networks/LSQ_layer.py:186-207, BEV/Networks/LSQ_layer.py:198-226).  Batch <= 64 rows against 4..256 output
//    features: weight-streaming GEMVs (16.8 MB of fp32 weights for the largest; HBM-bound, microseconds), fp32 with fp32
//    accumulation in a FIXED order (deterministic), optional fused ReLU.
//  * lf_seg_maps: the segmentation-mode fit input of Net.forward(end_to_end=False) -- arg-max over the class logits,
//    per-lane maps valued k where the arg-max is k, the masked top rows zeroed, and the BP tree's "prevent singular matrix"
//    overwrite of flagged lanes with map [0, 0] (BEV/Networks/LSQ_layer.py:302-308, BP/Networks/LSQ_layer.py:279-293,308-311)
//    -- one elementwise launch instead of argmax / compare / stack / index_put, and no host read of gt_line.sum().
#include "lf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NB = 8;      // batch rows per workgroup (forward, data gradient)
constexpr int OB = 4;      // output features per thread (weight gradient)

__device__ __forceinline__ float block_sum_256(float v, float* sm) {      // sum over 256 threads, fixed order
    v = lf_wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// y[n][o] = act(b[o] + sum_k x[n][k] * w[o][k]); grid (O, ceil(N / NB)); the weight row is read once per workgroup
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int N, int K, int O,
                                                        int relu) {
    const int o = blockIdx.x, n0 = blockIdx.y * NB;
    float acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = 0.f;
    const float* wr = w + (long)o * K;
    if ((K & 3) == 0) {
        for (int k = threadIdx.x * 4; k < K; k += 1024) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (n0 + j < N) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)(n0 + j) * K + k);
                    acc[j] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc[j]))));
                }
            }
        }
    } else {
        for (int k = threadIdx.x; k < K; k += 256) {
            const float wv = wr[k];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (n0 + j < N) acc[j] = fmaf(x[(long)(n0 + j) * K + k], wv, acc[j]);
        }
    }
    __shared__ float sm[4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float s = block_sum_256(acc[j], sm);
        if (threadIdx.x == 0 && n0 + j < N) {
            float v = s + (b ? b[o] : 0.f);
            if (relu) v = fmaxf(v, 0.f);
            y[(long)(n0 + j) * O + o] = v;
        }
    }
}

// gym[n][o] = gy[n][o] * [y[n][o] > 0] (relu) -- applied on the fly below
// gx[n][k] = sum_o gym[n][o] * w[o][k]; grid (ceil(K / 1024), ceil(N / NB)); a thread owns 4 consecutive k
__global__ __launch_bounds__(256) void linear_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ yact,
                                                             const float* __restrict__ w, float* __restrict__ gx, int N, int K,
                                                             int O, int relu) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4, n0 = blockIdx.y * NB;
    extern __shared__ float gs[];                 // [NB][O] masked upstream gradient of this workgroup's rows
    for (int i = threadIdx.x; i < NB * O; i += 256) {
        const int j = i / O, o = i - j * O;
        float g = 0.f;
        if (n0 + j < N) {
            g = gy[(long)(n0 + j) * O + o];
            if (relu && !(yact[(long)(n0 + j) * O + o] > 0.f)) g = 0.f;
        }
        gs[i] = g;
    }
    __syncthreads();
    if (k >= K) return;
    const bool vec = (K & 3) == 0;
    f32x4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int o = 0; o < O; ++o) {
        f32x4 wv;
        if (vec) wv = *reinterpret_cast<const f32x4*>(w + (long)o * K + k);
        else {
            wv.x = w[(long)o * K + k]; wv.y = k + 1 < K ? w[(long)o * K + k + 1] : 0.f;
            wv.z = k + 2 < K ? w[(long)o * K + k + 2] : 0.f; wv.w = k + 3 < K ? w[(long)o * K + k + 3] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] += gs[j * O + o] * wv;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (n0 + j >= N) continue;
        float* d = gx + (long)(n0 + j) * K + k;
        if (vec) *reinterpret_cast<f32x4*>(d) = acc[j];
        else {
            d[0] = acc[j].x;
            if (k + 1 < K) d[1] = acc[j].y;
            if (k + 2 < K) d[2] = acc[j].z;
            if (k + 3 < K) d[3] = acc[j].w;
        }
    }
}

// gw[o][k] = sum_n gym[n][o] * x[n][k]; gb[o] = sum_n gym[n][o]; grid (ceil(K / 1024), ceil(O / OB)); batch order fixed
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ yact,
                                                               const float* __restrict__ x, float* __restrict__ gw,
                                                               float* __restrict__ gb, int N, int K, int O, int relu) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4, o0 = blockIdx.y * OB;
    extern __shared__ float gs[];                 // [N][OB]
    for (int i = threadIdx.x; i < N * OB; i += 256) {
        const int n = i / OB, j = i - n * OB;
        float g = 0.f;
        if (o0 + j < O) {
            g = gy[(long)n * O + o0 + j];
            if (relu && !(yact[(long)n * O + o0 + j] > 0.f)) g = 0.f;
        }
        gs[i] = g;
    }
    __syncthreads();
    if (gb && blockIdx.x == 0 && threadIdx.x < OB && o0 + threadIdx.x < O) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += gs[n * OB + threadIdx.x];
        gb[o0 + threadIdx.x] = s;
    }
    if (k >= K) return;
    const bool vec = (K & 3) == 0;
    f32x4 acc[OB];
#pragma unroll
    for (int j = 0; j < OB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        f32x4 xv;
        if (vec) xv = *reinterpret_cast<const f32x4*>(x + (long)n * K + k);
        else {
            xv.x = x[(long)n * K + k]; xv.y = k + 1 < K ? x[(long)n * K + k + 1] : 0.f;
            xv.z = k + 2 < K ? x[(long)n * K + k + 2] : 0.f; xv.w = k + 3 < K ? x[(long)n * K + k + 3] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < OB; ++j) acc[j] += gs[n * OB + j] * xv;
    }
#pragma unroll
    for (int j = 0; j < OB; ++j) {
        if (o0 + j >= O) continue;
        float* d = gw + (long)(o0 + j) * K + k;
        if (vec) *reinterpret_cast<f32x4*>(d) = acc[j];
        else {
            d[0] = acc[j].x;
            if (k + 1 < K) d[1] = acc[j].y;
            if (k + 2 < K) d[2] = acc[j].z;
            if (k + 3 < K) d[3] = acc[j].w;
        }
    }
}

// maps[n][l][p] = (argmax_c logits[n][c][p] == l + 1) ? l + 1 : 0, rows < zero_rows zeroed; a lane flagged in gt_line takes the
// value map [0][0] has at p (computed here from image 0's logits, AFTER the row mask, as the reference's statement order gives)
__global__ __launch_bounds__(256) void seg_maps_kernel(const float* __restrict__ logits, const float* __restrict__ flags,
                                                      float* __restrict__ maps, int N, int C, int L, long P, long first) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= P) return;
    auto argmax_at = [&](int img) {
        const float* s = logits + (long)img * C * P + p;
        float best = s[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = s[(long)c * P];
            if (v > best || (v != v && best == best)) { best = v; bi = c; }      // first maximum; NaN counts as maximal (torch.argmax)
        }
        return bi;
    };
    const bool live = p >= first;
    const int am = live ? argmax_at(n) : -1;
    int am0 = -2;
    for (int l = 0; l < L; ++l) {
        float v = (live && am == l + 1) ? (float)(l + 1) : 0.f;
        if (flags && flags[n * L + l] != 0.f) {
            if (am0 == -2) am0 = live ? argmax_at(0) : -1;
            v = (live && am0 == 1) ? 1.f : 0.f;
        }
        maps[((long)n * L + l) * P + p] = v;
    }
}

}  // namespace

extern "C" {

// nn.Linear forward (+ optional ReLU): x (N,K), w (O,K), b (O) or NULL -> y (N,O).  fp32 row-major device tensors.
// Replaces F.linear / F.relu(F.linear) of Classification.forward (BP/Networks/LSQ_layer.py:199-207).
int lf_linear_fwd(const float* x, const float* w, const float* b, float* y, int N, int K, int O, int relu, void* stream) {
    LF_REQUIRE(x && w && y && N > 0 && K > 0 && O > 0, "lf_linear_fwd: bad arguments (N=%d K=%d O=%d)", N, K, O);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(O, lf_cdiv(N, NB)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, N, K, O, relu);
    LF_CHECK_LAUNCH("linear_fwd");
    return 0;
}

// Its backward: gy (N,O) [, y (N,O): the forward's output, needed when relu] -> gx (N,K) or NULL, gw (O,K) or NULL, gb (O) or
// NULL.  Written, not accumulated.
int lf_linear_bwd(const float* x, const float* w, const float* y, const float* gy, float* gx, float* gw, float* gb, int N, int K,
                  int O, int relu, void* stream) {
    LF_REQUIRE(x && w && gy && (y || !relu) && N > 0 && K > 0 && O > 0, "lf_linear_bwd: bad arguments (N=%d K=%d O=%d)", N, K, O);
    LF_REQUIRE((size_t)NB * O * sizeof(float) <= 48 * 1024 && (size_t)N * OB * sizeof(float) <= 48 * 1024,
               "lf_linear_bwd: N=%d / O=%d beyond the staged-gradient limits", N, O);
    LF_REQUIRE(gw || !gb, "lf_linear_bwd: the bias gradient rides in the weight-gradient launch");
    hipStream_t st = (hipStream_t)stream;
    if (gx) {
        hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(lf_cdiv(K, 1024), lf_cdiv(N, NB)), dim3(256), NB * O * sizeof(float), st, gy, y,
                           w, gx, N, K, O, relu);
        LF_CHECK_LAUNCH("linear_bwd_data");
    }
    if (gw) {
        hipLaunchKernelGGL(linear_bwd_weight_kernel, dim3(lf_cdiv(K, 1024), lf_cdiv(O, OB)), dim3(256), N * OB * sizeof(float), st, gy,
                           y, x, gw, gb, N, K, O, relu);
        LF_CHECK_LAUNCH("linear_bwd_weight");
    }
    return 0;
}

// Segmentation-mode fit input: logits (N,C,H,W) NCHW -> maps (N,L,H,W), L lanes (normally C - 1: 2 or 4); zero_rows masked rows;
// gt_line (N,L) fp32 device flags or NULL (BP only: "prevent singular matrix").  See the file header.
int lf_seg_maps(const float* logits, const float* gt_line, float* maps, int N, int C, int L, int H, int W, int zero_rows,
                void* stream) {
    // (L >= C is allowed: a net built with out_channels == lanes and called with end_to_end=False -- the lanes beyond C - 1 never win
    // the arg-max and come out all zero, as in the reference's statement sequence)
    LF_REQUIRE(logits && maps && N > 0 && C >= 1 && L >= 1 && H > 0 && W > 0 && zero_rows >= 0 && zero_rows <= H,
               "lf_seg_maps: bad arguments (N=%d C=%d L=%d H=%d W=%d zero_rows=%d)", N, C, L, H, W, zero_rows);
    const long P = (long)H * W;
    hipLaunchKernelGGL(seg_maps_kernel, dim3(lf_cdiv(P, 256), N), dim3(256), 0, (hipStream_t)stream, logits, gt_line, maps, N, C, L,
                       P, (long)zero_rows * W);
    LF_CHECK_LAUNCH("seg_maps");
    return 0;
}

}  // extern "C"
