// Chain of [Conv2d(k x k, stride 1, pad k/2, bias) -> BatchNorm2d -> ReLU] blocks on an NHWC fp32 tensor,
// forward and backward, on the tap-GEMM kernels of lf_conv.hip.
//
// Replaces the convolutional trunk of `Classification` (BP/Networks/LSQ_layer.py:150-191: conv1..conv4 with
// their BatchNorms, :178-181) that the `--clas` line-type / horizon heads run on the shared encoder output.
// The same fusions as the backbone apply: BatchNorm statistics come from the producing conv's epilogue, the
// BN+ReLU of block i is applied while block i+1 loads its operand (so post-activation tensors are never
// written, except the last), and in the backward pass the data-gradient of block i+1 applies the ReLU mask
// of block i and accumulates its BatchNorm-backward sums.
#include <stdlib.h>

#include <vector>

#include "lf_conv.h"
#include "lf_eltwise.h"
#include "lf_plan.h"

#define LF_CHAIN_MAX 8

struct lf_convchain_plan {
    int N, H, W, L;
    int C[LF_CHAIN_MAX + 1], k[LF_CHAIN_MAX];
    LfTapGeom fwd[LF_CHAIN_MAX], dg[LF_CHAIN_MAX];
    int pk_fwd[LF_CHAIN_MAX], pk_dg[LF_CHAIN_MAX];
    std::vector<LfPackEntry> packs;
    long packed_floats;
    long z[LF_CHAIN_MAX];                               // pre-BN conv outputs
    long sc[LF_CHAIN_MAX], sh[LF_CHAIN_MAX], asc[LF_CHAIN_MAX], ash[LF_CHAIN_MAX], c1[LF_CHAIN_MAX], c2[LF_CHAIN_MAX];
    long off_entries, off_packed, off_stat, stat_floats, off_wpart, wpart_floats, off_bpart, bpart_floats;
    long off_gA, off_gB, gbuf_floats, total_floats;
};

namespace {

int add_pack(lf_convchain_plan* P, int param, int Kc, int Nc, long sk, long sn, const LfTapGeom& g, const int* tapidx) {
    LfPackEntry e;
    memset(&e, 0, sizeof(e));
    e.param = param; e.Kc = Kc; e.Nc = Nc; e.ntaps = g.ntaps; e.sk = sk; e.sn = sn;
    for (int t = 0; t < g.ntaps; ++t) e.tapidx[t] = tapidx[t];
    e.dst_off = P->packed_floats;
    e.dst16_off = 0;                       // the heads stay on the fp32 matrix-core path
    P->packed_floats += (long)g.ntaps * Kc * Nc;
    P->packs.push_back(e);
    return (int)P->packs.size() - 1;
}


// ---- pooling + flatten in front of the fully connected layers (LSQ_layer.py:183-187) ---------------------
// mode 0 ('line'):    MaxPool2d(2, 2)        NHWC (N,H,W,C) -> NCHW-flat (N, C*(H/2)*(W/2))
// mode 1 ('horizon'): AvgPool2d((1, W))      NHWC (N,H,W,C) -> NCHW-flat (N, C*H)
__global__ __launch_bounds__(256) void poolflat_max_fwd_kernel(const float* __restrict__ y, int N, int H, int W, int C,
                                                              float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * C * Ho * Wo;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int j = (int)(u % Wo);
        long r = u / Wo;
        const int i = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C), n = (int)(r / C);
        const float* b = y + (((long)n * H + 2 * i) * W + 2 * j) * C + c;
        out[u] = fmaxf(fmaxf(b[0], b[C]), fmaxf(b[(long)W * C], b[(long)W * C + C]));
    }
}
__global__ __launch_bounds__(256) void poolflat_max_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                              int N, int H, int W, int C, float* __restrict__ gy) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int c = (int)(u % C);
        long r = u / C;
        const int j = (int)(r % Wo);
        r /= Wo;
        const int i = (int)(r % Ho), n = (int)(r / Ho);
        const long base = (((long)n * H + 2 * i) * W + 2 * j) * C + c;
        const long o[4] = {0, C, (long)W * C, (long)W * C + C};
        int arg = 0;
        float m = y[base];
#pragma unroll
        for (int e = 1; e < 4; ++e) {      // first maximum in window scan order wins, as ATen's max_pool2d
            const float v = y[base + o[e]];
            if (v > m) { m = v; arg = e; }
        }
        const float gv = g[(((long)n * C + c) * Ho + i) * Wo + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) gy[base + o[e]] = e == arg ? gv : 0.f;
    }
}
__global__ __launch_bounds__(256) void poolflat_avg_fwd_kernel(const float* __restrict__ y, int N, int H, int W, int C,
                                                              float* __restrict__ out) {
    const long total = (long)N * H * C;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int c = (int)(u % C);
        const long r = u / C;
        const int i = (int)(r % H), n = (int)(r / H);
        const float* b = y + ((long)n * H + i) * W * C + c;
        float s = 0.f;
        for (int j = 0; j < W; ++j) s += b[(long)j * C];
        out[((long)n * C + c) * H + i] = s / (float)W;
    }
}
__global__ __launch_bounds__(256) void poolflat_avg_bwd_kernel(const float* __restrict__ g, int N, int H, int W, int C,
                                                              float* __restrict__ gy) {
    const long total = (long)N * H * W * C;
    const float inv = 1.f / (float)W;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int c = (int)(u % C);
        const long r = u / C / W;
        const int i = (int)(r % H), n = (int)(r / H);
        gy[u] = g[((long)n * C + c) * H + i] * inv;
    }
}

int poolflat_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" {

// channels: L+1 entries (input, then each block's output; multiples of 16); ksize: L entries (1 or 3).
// Parameters are passed 4 per block in module order: conv.weight, conv.bias, bn.weight, bn.bias.
lf_convchain_plan* lf_convchain_plan_create(int N, int H, int W, int nlayers, const int* channels, const int* ksize) {
    if (N < 1 || H < 1 || W < 1 || nlayers < 1 || nlayers > LF_CHAIN_MAX || !channels || !ksize ||
        (long)N * H * W >= (1l << 31)) {
        lf_fail("lf_convchain_plan_create: bad arguments");
        return nullptr;
    }
    for (int i = 0; i <= nlayers; ++i)
        if (channels[i] < 16 || channels[i] % 16 != 0) { lf_fail("lf_convchain_plan_create: channels must be multiples of 16"); return nullptr; }
    for (int i = 0; i < nlayers; ++i)
        if (ksize[i] != 1 && ksize[i] != 3) { lf_fail("lf_convchain_plan_create: kernel size must be 1 or 3"); return nullptr; }
    lf_convchain_plan* P = new lf_convchain_plan();
    P->N = N; P->H = H; P->W = W; P->L = nlayers; P->packed_floats = 0;
    P->stat_floats = 0; P->wpart_floats = 0; P->bpart_floats = 0;
    LfBump ws;
    const long npix = (long)N * H * W;
    int cmax = 0;
    for (int i = 0; i <= nlayers; ++i) { P->C[i] = channels[i]; cmax = channels[i] > cmax ? channels[i] : cmax; }
    for (int i = 0; i < nlayers; ++i) {
        const int k = ksize[i], Ci = channels[i], Co = channels[i + 1], kk = k * k;
        P->k[i] = k;
        LfTapGeom g = lf_base_geom(N, H, W, H, W, Ci, H, W, Co, Ci, Co);
        LfTapGeom d = lf_base_geom(N, H, W, H, W, Co, H, W, Ci, Co, Ci);
        int idx_f[9], idx_d[9];
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                const int t = a * k + b;
                g.tdh[t] = d.tdh[t] = a - k / 2;
                g.tdw[t] = d.tdw[t] = b - k / 2;
                idx_f[t] = t;
                idx_d[t] = kk - 1 - t;         // data gradient = correlation with the flipped kernel
            }
        g.ntaps = d.ntaps = kk;
        P->fwd[i] = g; P->dg[i] = d;
        // Conv2d weight (Co, Ci, k, k)
        P->pk_fwd[i] = add_pack(P, 4 * i, Ci, Co, /*sk (ci)*/ kk, /*sn (co)*/ (long)Ci * kk, g, idx_f);
        P->pk_dg[i] = add_pack(P, 4 * i, Co, Ci, /*sk (co)*/ (long)Ci * kk, /*sn (ci)*/ kk, d, idx_d);
        P->z[i] = ws.take(npix * Co);
        P->sc[i] = ws.take(Co); P->sh[i] = ws.take(Co); P->asc[i] = ws.take(Co); P->ash[i] = ws.take(Co);
        P->c1[i] = ws.take(Co); P->c2[i] = ws.take(Co);
        P->stat_floats = lf_maxl(P->stat_floats, (long)lf_tapgemm_stat_rows(g) * 2 * Co);
        P->stat_floats = lf_maxl(P->stat_floats, (long)lf_tapgemm_stat_rows(d) * 2 * Ci);
        P->wpart_floats = lf_maxl(P->wpart_floats, (long)lf_tapwgrad_splits(g) * kk * Ci * Co);
        P->bpart_floats = lf_maxl(P->bpart_floats, (long)lf_tapwgrad_splits_bound(g, 0) * Co);      // (fp32 tensors: the chain has no bf16 mode)
    }
    P->stat_floats = lf_maxl(P->stat_floats, (long)lf_bn_bwd_reduce_rows(npix) * 2 * cmax);
    P->off_entries = ws.take((long)(P->packs.size() * sizeof(LfPackEntry) + 3) / 4);
    P->off_packed = ws.take(P->packed_floats);
    P->stat_floats += 2L * cmax * 4;                 // (channel-major rows: leading dimensions are rounded up to 4)
    P->off_stat = ws.take((P->stat_floats + 3) / 4 * 4);
    P->off_wpart = ws.take(P->wpart_floats);
    P->off_bpart = ws.take(P->bpart_floats);
    P->gbuf_floats = npix * cmax;
    P->off_gA = ws.take(P->gbuf_floats);
    P->off_gB = ws.take(P->gbuf_floats);
    P->total_floats = ws.cur;
    return P;
}

void lf_convchain_plan_destroy(lf_convchain_plan* P) { delete P; }
size_t lf_convchain_workspace_bytes(const lf_convchain_plan* P) { return (size_t)P->total_floats * sizeof(float); }

// x (N,H,W,C0) NHWC; params_host / params_dev: 4*L device pointers (host array / device array);
// running_host: 2*L device pointers (running_mean, running_var per block); y (N,H,W,C_L) NHWC = the last
// block's post-ReLU output.  The workspace keeps what lf_convchain_backward needs.
int lf_convchain_forward(const lf_convchain_plan* P, const float* x, const float* const* params_host,
                         const float* const* params_dev, float* const* running_host, int training, float momentum,
                         float eps, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    LF_REQUIRE(P && x && params_host && params_dev && running_host && y && workspace, "lf_convchain_forward: null pointer");
    LF_REQUIRE(workspace_bytes >= lf_convchain_workspace_bytes(P), "lf_convchain_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    LfPackEntry* ent = reinterpret_cast<LfPackEntry*>(ws + P->off_entries);
    if (hipMemcpyAsync(ent, P->packs.data(), P->packs.size() * sizeof(LfPackEntry), hipMemcpyHostToDevice, st) != hipSuccess)
        return lf_fail("lf_convchain_forward: upload of the pack table failed");
    LF_TRY(lf_pack_weights_launch(ent, (int)P->packs.size(), params_dev, ws + P->off_packed, st));
    const long npix = (long)P->N * P->H * P->W;
    float* stat = ws + P->off_stat;
    for (int i = 0; i < P->L; ++i) {
        LfTapArgs a = lf_no_args();
        a.src = i == 0 ? x : ws + P->z[i - 1];
        a.dst = ws + P->z[i];
        a.wp = ws + P->off_packed + P->packs[P->pk_fwd[i]].dst_off;
        a.bias = params_host[4 * i + 1];
        a.stats = stat; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(P->fwd[i]));
        int pro = LF_PRO_NONE;
        if (i > 0) { a.pro_sc = ws + P->sc[i - 1]; a.pro_sh = ws + P->sh[i - 1]; pro = LF_PRO_BNRELU; }
        LF_TRY(lf_tapgemm_launch(P->fwd[i], a, pro, training ? LF_EPI_STATS_SQ : 0, st));
        const int srows = lf_tapgemm_stat_rows_for(P->fwd[i], a);
        LfStatPart part = lf_stat_part_tiles(stat, srows, a.stats_ld, P->C[i + 1], 0, srows, npix);      // centred rows (LfStatPart)
        LF_TRY(lf_bn_finalize_fwd(&part, 1, P->C[i + 1], (double)npix, params_host[4 * i + 2], params_host[4 * i + 3],
                                  running_host[2 * i], running_host[2 * i + 1], momentum, eps, training, ws + P->sc[i],
                                  ws + P->sh[i], ws + P->asc[i], ws + P->ash[i], st));
    }
    const int l = P->L - 1;
    return lf_bn_act(ws + P->z[l], ws + P->sc[l], ws + P->sh[l], nullptr, nullptr, y, npix, P->C[P->L], (long)P->H * P->W, 0, st);
}

// Backward of the forward that last used `workspace`.  gy (N,H,W,C_L) NHWC; grads_host: 4*L device pointers
// (written, not accumulated; all must be non-null); gx (N,H,W,C0) NHWC or NULL when the input needs no gradient;
// training = the mode of that forward (0: BatchNorm used its running statistics, its backward is the affine map's).
int lf_convchain_backward(const lf_convchain_plan* P, const float* x, const float* y, const float* gy,
                          const float* const* params_host, float* const* grads_host, float* gx, int training, void* workspace,
                          size_t workspace_bytes, void* stream) {
    LF_REQUIRE(P && x && y && gy && params_host && grads_host && workspace, "lf_convchain_backward: null pointer");
    LF_REQUIRE(workspace_bytes >= lf_convchain_workspace_bytes(P), "lf_convchain_backward: workspace too small");
    for (int i = 0; i < 4 * P->L; ++i) LF_REQUIRE(grads_host[i], "lf_convchain_backward: gradient pointer %d is null", i);
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const long npix = (long)P->N * P->H * P->W, ppi = (long)P->H * P->W;
    float* stat = ws + P->off_stat;
    float *A = ws + P->off_gA, *B = ws + P->off_gB;
    // last block: BatchNorm + ReLU backward from the saved output
    int l = P->L - 1;
    const int rrows = lf_bn_bwd_reduce_rows(npix);
    LF_TRY(lf_bn_bwd_reduce(gy, y, ws + P->z[l], ws + P->asc[l], ws + P->ash[l], nullptr, stat, lf_stat_ld(rrows), npix, P->C[l + 1], ppi, 0, st));
    LfStatPart rp = {stat, rrows, P->C[l + 1], 0, lf_stat_ld(rrows)};
    LF_TRY(lf_bn_bwd_finalize(&rp, 1, P->C[l + 1], (double)npix, ws + P->asc[l], ws + P->ash[l], ws + P->c1[l], ws + P->c2[l],
                              grads_host[4 * l + 2], grads_host[4 * l + 3], training, st));
    LF_TRY(lf_bn_bwd_apply(gy, y, ws + P->z[l], ws + P->asc[l], ws + P->ash[l], params_host[4 * l + 2], ws + P->c1[l],
                           ws + P->c2[l], nullptr, A, nullptr, npix, P->C[l + 1], ppi, 0, st));
    float *gz = A, *other = B;      // gz = d loss / d z_i
    for (int i = l; i >= 0; --i) {
        // weight + bias gradient: input a_{i-1} = relu(bn_{i-1}(z_{i-1})) recomputed on the operand load
        LfWgradArgs wa;
        wa.x = i == 0 ? x : ws + P->z[i - 1];
        wa.g = gz;
        wa.s16 = 0;
        wa.split = 0;                   // the heads stay on the fp32 matrix cores
        wa.pro_sc = i == 0 ? nullptr : ws + P->sc[i - 1];
        wa.pro_sh = i == 0 ? nullptr : ws + P->sh[i - 1];
        wa.partial = ws + P->off_wpart;
        wa.bias_partial = ws + P->off_bpart;
        LF_TRY(lf_tapwgrad_launch(P->fwd[i], wa, i == 0 ? LF_PRO_NONE : LF_PRO_BNRELU, st));
        const LfPackEntry& e = P->packs[P->pk_fwd[i]];
        const int nsplit = lf_tapwgrad_splits_for(P->fwd[i], wa, i == 0 ? LF_PRO_NONE : LF_PRO_BNRELU);
        LF_TRY(lf_wgrad_reduce_launch(wa.partial, nsplit, P->fwd[i].ntaps, P->fwd[i].Cs, P->fwd[i].Cd,
                                      grads_host[4 * i], e.sk, e.sn, e.tapidx, wa.bias_partial,
                                      nsplit, grads_host[4 * i + 1], 0, st));
        // data gradient
        LfTapArgs a = lf_no_args();
        a.src = gz;
        a.wp = ws + P->off_packed + P->packs[P->pk_dg[i]].dst_off;
        if (i == 0) {
            if (gx) { a.dst = gx; LF_TRY(lf_tapgemm_launch(P->dg[i], a, LF_PRO_NONE, 0, st)); }
            break;
        }
        const int j = i - 1;
        a.dst = other;
        a.aux = ws + P->z[j]; a.msc = ws + P->sc[j]; a.msh = ws + P->sh[j]; a.asc = ws + P->asc[j]; a.ash = ws + P->ash[j];
        a.stats = stat; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(P->dg[i]));
        LF_TRY(lf_tapgemm_launch(P->dg[i], a, LF_PRO_NONE, LF_EPI_MASKBN | LF_EPI_STATS_XHAT, st));
        LfStatPart sp = {stat, lf_tapgemm_stat_rows_for(P->dg[i], a), P->C[i], 0, a.stats_ld};
        LF_TRY(lf_bn_bwd_finalize(&sp, 1, P->C[i], (double)npix, ws + P->asc[j], ws + P->ash[j], ws + P->c1[j], ws + P->c2[j],
                                  grads_host[4 * j + 2], grads_host[4 * j + 3], training, st));
        LF_TRY(lf_bn_bwd_apply(other, nullptr, ws + P->z[j], ws + P->asc[j], ws + P->ash[j], params_host[4 * j + 2],
                               ws + P->c1[j], ws + P->c2[j], nullptr, gz, nullptr, npix, P->C[i], ppi, 0, st));
        // gz now holds d loss / d z_{i-1} (written over the consumed gradient), `other` is scratch again
    }
    return 0;
}

int lf_poolflat_fwd(const float* y, int N, int H, int W, int C, int mode, float* out, void* stream) {
    LF_REQUIRE(y && out && N > 0 && C > 0, "lf_poolflat_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {
        LF_REQUIRE(H % 2 == 0 && W % 2 == 0, "lf_poolflat_fwd: max-pool needs even H, W");
        hipLaunchKernelGGL(poolflat_max_fwd_kernel, dim3(poolflat_grid((long)N * C * (H / 2) * (W / 2))), dim3(256), 0, st, y, N, H, W, C, out);
    } else {
        hipLaunchKernelGGL(poolflat_avg_fwd_kernel, dim3(poolflat_grid((long)N * H * C)), dim3(256), 0, st, y, N, H, W, C, out);
    }
    LF_CHECK_LAUNCH("lf_poolflat_fwd");
    return 0;
}

// gy (N,H,W,C) NHWC is fully written (zeros at the non-maximal positions)
int lf_poolflat_bwd(const float* y, const float* g, int N, int H, int W, int C, int mode, float* gy, void* stream) {
    LF_REQUIRE(y && g && gy && N > 0 && C > 0, "lf_poolflat_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {
        LF_REQUIRE(H % 2 == 0 && W % 2 == 0, "lf_poolflat_bwd: max-pool needs even H, W");
        hipLaunchKernelGGL(poolflat_max_bwd_kernel, dim3(poolflat_grid((long)N * (H / 2) * (W / 2) * C)), dim3(256), 0, st, y, g, N, H, W, C, gy);
    } else {
        hipLaunchKernelGGL(poolflat_avg_bwd_kernel, dim3(poolflat_grid((long)N * H * W * C)), dim3(256), 0, st, g, N, H, W, C, gy);
    }
    LF_CHECK_LAUNCH("lf_poolflat_bwd");
    return 0;
}

}  // extern "C"
