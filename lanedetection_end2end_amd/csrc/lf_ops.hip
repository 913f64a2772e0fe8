// Kernel-level C-ABI entry points: one factorised 1-D convolution (forward, data gradient, weight
// gradient) on NHWC fp32 tensors, outside the ERFNet plan.  Used by the kernel-level parity tests
// and by tools/kbench.py (roofline micro-benchmarks, kernel A/B switches).
// Replaces nn.Conv2d(C, C, (3,1)|(1,3), padding=d, dilation=d) of non_bottleneck_1d (ERFNet.py:29-37).
#include "lf_conv.h"
#include "lf_debug.h"

namespace {

__global__ __launch_bounds__(256) void pack_one_kernel(const float* __restrict__ w, float* __restrict__ dst, int Kc, int Nc,
                                                      int ntaps, long sk, long sn, int flip) {
    const long total = (long)ntaps * Kc * Nc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k4 = (int)(i & 3);
        long r = i >> 2;
        const int n = (int)(r % Nc);
        r /= Nc;
        const int kb = (int)(r % (Kc >> 2));
        const int t = (int)(r / (Kc >> 2));
        dst[i] = w[(kb * 4 + k4) * sk + n * sn + (flip ? ntaps - 1 - t : t)];
    }
}

// bf16 operand order of tapgemm_bf16_kernel: [tap][ceil(Kc/32)*4][Nc][8], zero beyond Kc
__global__ __launch_bounds__(256) void pack_one_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int Kc, int Nc,
                                                           int ntaps, long sk, long sn, int flip) {
    const int kb_per_tap = ((Kc + 31) >> 5) * 4;
    const long total = (long)ntaps * kb_per_tap * Nc * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k8 = (int)(i & 7);
        long r = i >> 3;
        const int n = (int)(r % Nc);
        r /= Nc;
        const int kb = (int)(r % kb_per_tap);
        const int t = (int)(r / kb_per_tap);
        const int k = kb * 8 + k8;
        dst[i] = (__bf16)(k < Kc ? w[k * sk + n * sn + (flip ? ntaps - 1 - t : t)] : 0.f);
    }
}

// split operand order of tapgemm_split_kernel: [tap][Kc/8][Nc][3][8] bf16 (pieces h, m, l; h + m + l == w exactly)
__global__ __launch_bounds__(256) void pack_one_split_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int Kc, int Nc,
                                                            int ntaps, long sk, long sn, int flip) {
    const int kb_per_tap = Kc >> 3;
    const long total = (long)ntaps * kb_per_tap * Nc * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k8 = (int)(i & 7);
        const long row = i >> 3;
        long r = row;
        const int n = (int)(r % Nc);
        r /= Nc;
        const int kb = (int)(r % kb_per_tap);
        const int t = (int)(r / kb_per_tap);
        const float v = w[(kb * 8 + k8) * sk + n * sn + (flip ? ntaps - 1 - t : t)];
        const __bf16 vh = (__bf16)v;
        const float r1 = v - (float)vh;
        const __bf16 vm = (__bf16)r1;
        dst[row * 24 + k8] = vh; dst[row * 24 + 8 + k8] = vm; dst[row * 24 + 16 + k8] = (__bf16)(r1 - (float)vm);
    }
}

int g_ops_bf16 = 0;     // 0 fp32 cores, 2 bf16 cores on bf16 tensors, 9 fp32 from 9-term split operands on the bf16 cores

// pack into scratch in the order the selected kernel wants; returns the LfTapArgs weight fields
void pack_conv1d(LfTapArgs& a, const float* w, float* scratch, int C, long sk, long sn, int flip, hipStream_t st) {
    if (g_ops_bf16 == 9) {
        hipLaunchKernelGGL(pack_one_kernel, dim3(64), dim3(256), 0, st, w, scratch, C, C, 3, sk, sn, flip);
        if (C % 32 == 0) {
            hipLaunchKernelGGL(pack_one_split_kernel, dim3(64), dim3(256), 0, st, w, reinterpret_cast<__bf16*>(scratch + 3L * C * C),
                               C, C, 3, sk, sn, flip);
            a.split = g_ops_bf16;
            a.wp48 = scratch + 3L * C * C;
        }
    } else if (g_ops_bf16 == 2) {
        hipLaunchKernelGGL(pack_one_bf16_kernel, dim3(64), dim3(256), 0, st, w, reinterpret_cast<__bf16*>(scratch), C, C, 3, sk, sn, flip);
        a.wp16 = scratch;
        a.s16 = 1;
    } else {
        hipLaunchKernelGGL(pack_one_kernel, dim3(64), dim3(256), 0, st, w, scratch, C, C, 3, sk, sn, flip);
    }
    a.wp = scratch;
}

LfTapGeom conv1d_geom(int N, int H, int W, int C, int axis, int d) {
    LfTapGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Hl = H; g.Wl = W; g.Hs = H; g.Ws = W; g.s_pix = C; g.ssh = 1; g.ssw = 1;
    g.Hd = H; g.Wd = W; g.d_pix = C; g.dsh = 1; g.dsw = 1; g.Cs = C; g.Cd = C; g.ntaps = 3;
    for (int t = 0; t < 3; ++t) { g.tdh[t] = axis == 0 ? (t - 1) * d : 0; g.tdw[t] = axis == 1 ? (t - 1) * d : 0; }
    return g;
}

}  // namespace

extern "C" {

void lf_debug_set_split_any_size(int v) { lf_tapgemm_set_split_any_size(v); }
void lf_debug_set_bf16_lds(int v) { lf_tapgemm_set_bf16_lds(v); }
// precision mode of the kernel-level conv1d calls below (tests, kbench): 0 fp32, 2 bf16 matrix cores on bf16 tensors (x, y, gx, gy,
// mask_src then hold bf16 elements; w, bias, gw, gb stay fp32), 9 fp32 from 9-term split operands; anything else (the removed
// modes 1 and 6) selects 0
void lf_debug_set_ops_precision(int mode) { g_ops_bf16 = (mode == 2 || mode == 9) ? mode : 0; }

void lf_debug_set_wgrad_ro(int mode, int cap64, int cap128) { lf_tapwgrad_ro_set(mode, cap64, cap128); }

// same as lf_conv1d_fwd with per-wave phase timestamps: dbg receives 8 uint64 per wave
// (start, tap table built, main loop done, stores retired); waves = ceil(N*H*W/256)*4*(C/64)
int lf_debug_conv1d_fwd_phases(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C,
                               int axis, int dilation, float* scratch, unsigned long long* dbg, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfTapArgs a;
    memset(&a, 0, sizeof(a));
    pack_conv1d(a, w, scratch, C, 3L, 3L * C, 0, st);
    a.src = x; a.bias = bias; a.dst = y; a.dbg = dbg;
    return lf_tapgemm_launch(g, a, LF_PRO_NONE, 0, st);
}

int lf_debug_conv1d_fwd_pro(const float* x, const float* w, const float* bias, const float* sc, const float* sh, float* y, int N, int H,
                            int W, int C, int axis, int dilation, float* scratch, void* stream) {
    LF_REQUIRE(x && w && y && sc && sh && scratch, "lf_debug_conv1d_fwd_pro: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfTapArgs a;
    memset(&a, 0, sizeof(a));
    pack_conv1d(a, w, scratch, C, 3L, 3L * C, 0, st);
    a.src = x; a.bias = bias; a.dst = y; a.pro_sc = sc; a.pro_sh = sh;
    return lf_tapgemm_launch(g, a, LF_PRO_BNRELU, LF_EPI_RELU, st);
}

// the weight gradient of lf_conv1d_bwd_weight (no reduction) with per-wave phase timestamps; returns the number of waves
int lf_debug_conv1d_wgrad_phases(const float* x, const float* gy, int N, int H, int W, int C, int axis, int dilation,
                                 float* scratch, unsigned long long* dbg, void* stream) {
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfWgradArgs a;
    a.x = x; a.g = gy; a.pro_sc = nullptr; a.pro_sh = nullptr; a.s16 = 0; a.split = 0;
    a.partial = scratch; a.bias_partial = nullptr; a.dbg = dbg;
    const int rc = lf_tapwgrad_launch(g, a, LF_PRO_NONE, (hipStream_t)stream);
    return rc ? -1 : lf_tapwgrad_splits_for(g, a, LF_PRO_NONE) * 3 * (C / 64) * (C / 64) * 4;
}

// scratch floats needed by the three calls below (packed weights / split-K partials)
long lf_conv1d_scratch_floats(int N, int H, int W, int C) {
    const LfTapGeom g = conv1d_geom(N, H, W, C, 0, 1);
    long a = 8L * C * C;        // fp32-packed weights + their 3-piece bf16 split (4.5 C^2 floats)
    const long rows = lf_tapwgrad_splits_bound(g, 1);          // either storage type
    long b = rows * 3 * C * C + rows * C;
    return a > b ? a : b;
}

// y = [relu](conv1d(x) + bias); x,y (N,H,W,C) NHWC; w in the nn.Conv2d layout (C,C,3) flattened; axis 0 = 3x1, 1 = 1x3
int lf_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int axis,
                  int dilation, int relu, float* scratch, void* stream) {
    LF_REQUIRE(x && w && y && scratch, "lf_conv1d_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfTapArgs a;
    memset(&a, 0, sizeof(a));
    pack_conv1d(a, w, scratch, C, 3L, 3L * C, 0, st);
    a.src = x; a.bias = bias; a.dst = y;
    return lf_tapgemm_launch(g, a, LF_PRO_NONE, relu ? LF_EPI_RELU : 0, st);
}

// gx = conv1d^T(gy) [* (mask_src > 0)]
int lf_conv1d_bwd_data(const float* gy, const float* w, const float* mask_src, float* gx, int N, int H, int W, int C,
                       int axis, int dilation, float* scratch, void* stream) {
    LF_REQUIRE(gy && w && gx && scratch, "lf_conv1d_bwd_data: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfTapArgs a;
    memset(&a, 0, sizeof(a));
    pack_conv1d(a, w, scratch, C, 3L * C, 3L, 1, st);
    a.src = gy; a.dst = gx; a.mask_src = mask_src;
    return lf_tapgemm_launch(g, a, LF_PRO_NONE, mask_src ? LF_EPI_MASK : 0, st);
}

int lf_debug_conv1d_bwd_data_epi3(const float* gy, const float* w, const float* mask_src, const float* add_src, const float* aux,
                                  float* gx, float* stats, int N, int H, int W, int C, int axis, int dilation, float* scratch, void* stream) {
    if (!(gy && w && mask_src && add_src && aux && gx && stats && scratch)) { lf_fail("lf_debug_conv1d_bwd_data_epi3: null pointer"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfTapArgs a;
    memset(&a, 0, sizeof(a));
    pack_conv1d(a, w, scratch, C, 3L * C, 3L, 1, st);
    a.src = gy; a.dst = gx; a.mask_src = mask_src; a.add_src = add_src; a.aux = aux; a.stats = stats;
    a.stats_ld = lf_tapgemm_stat_rows(g);        // the caller's buffer: [2][C][rows], rows = ceil(N * H * W / 256)
    if (lf_tapgemm_launch(g, a, LF_PRO_NONE, LF_EPI_ADD | LF_EPI_MASK | LF_EPI_STATS_XHAT, st)) return -1;
    return lf_tapgemm_stat_rows_for(g, a);
}

// one tap-GEMM launch with any of the network's epilogue flag sets (LF_EPI_*, lf_conv.h); transposed = 1: data-gradient weights.
// Tensors a flag does not name may be null.  Returns the number of statistics rows written (0 without a STATS flag), -1 on error.
int lf_debug_conv1d_epi(const float* src, const float* w, const float* bias, float* dst, int transposed, int epi, const float* mask_src,
                        const float* add_src, const float* aux, const float* msc, const float* msh, float* stats, int N, int H, int W, int C,
                        int axis, int dilation, float* scratch, void* stream) {
    if (!(src && w && dst && scratch)) { lf_fail("lf_debug_conv1d_epi: null pointer"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    LfTapArgs a;
    memset(&a, 0, sizeof(a));
    if (transposed) pack_conv1d(a, w, scratch, C, 3L * C, 3L, 1, st);
    else pack_conv1d(a, w, scratch, C, 3L, 3L * C, 0, st);
    a.src = src; a.dst = dst; a.bias = bias; a.mask_src = mask_src; a.add_src = add_src; a.aux = aux; a.msc = msc; a.msh = msh; a.stats = stats;
    a.stats_ld = lf_tapgemm_stat_rows(g);        // the caller's buffer: [2][C][rows], rows = ceil(N * H * W / 256)
    if (lf_tapgemm_launch(g, a, LF_PRO_NONE, epi, st)) return -1;
    return (epi & (LF_EPI_STATS_SQ | LF_EPI_STATS_XHAT)) ? lf_tapgemm_stat_rows_for(g, a) : 0;
}

namespace {
int conv1d_bwd_weight(const float* x, const float* gy, const float* sc, const float* sh, float* gw, float* gb, int N, int H, int W, int C,
                      int axis, int dilation, float* scratch, hipStream_t st) {
    const LfTapGeom g = conv1d_geom(N, H, W, C, axis, dilation);
    const int pro = sc ? LF_PRO_BNRELU : LF_PRO_NONE;
    LfWgradArgs a;
    a.x = x; a.g = gy; a.pro_sc = sc; a.pro_sh = sh; a.s16 = g_ops_bf16 == 2;
    a.split = 0;
    a.partial = scratch;
    a.bias_partial = gb ? scratch + (long)lf_tapwgrad_splits_bound(g, 1) * 3 * C * C : nullptr;
    int rc = lf_tapwgrad_launch(g, a, pro, st);
    if (rc) return rc;
    const int idx[3] = {0, 1, 2};
    const int nsplit = lf_tapwgrad_splits_for(g, a, pro);
    return lf_wgrad_reduce_launch(a.partial, nsplit, 3, C, C, gw, 3L, 3L * C, idx, a.bias_partial, nsplit, gb, 0, st);
}
}  // namespace

// gw (C,C,3) and gb (C) from x and gy
int lf_conv1d_bwd_weight(const float* x, const float* gy, float* gw, float* gb, int N, int H, int W, int C, int axis,
                         int dilation, float* scratch, void* stream) {
    LF_REQUIRE(x && gy && gw && scratch, "lf_conv1d_bwd_weight: null pointer");
    return conv1d_bwd_weight(x, gy, nullptr, nullptr, gw, gb, N, H, W, C, axis, dilation, scratch, (hipStream_t)stream);
}

int lf_debug_conv1d_wgrad_pro(const float* x, const float* gy, const float* sc, const float* sh, float* gw, float* gb, int N, int H, int W,
                              int C, int axis, int dilation, float* scratch, void* stream) {
    LF_REQUIRE(x && gy && sc && sh && gw && scratch, "lf_debug_conv1d_wgrad_pro: null pointer");
    return conv1d_bwd_weight(x, gy, sc, sh, gw, gb, N, H, W, C, axis, dilation, scratch, (hipStream_t)stream);
}

}  // extern "C"
