// Internal interface of the tap-GEMM convolution kernels (gfx950; fp32 MFMA 16x16x4 by default, bf16 MFMA 16x16x32 in the
// bf16 and split precision modes).
//
// Every convolution-like layer of the ERFNet backbone (1-D factorised 3x1 / 1x3 convs with
// dilation, 3x3 stride-2 convs, 3x3 stride-2 transposed convs by sub-pixel phase, and all of
// their data gradients) is ONE kernel family: for every logical pixel p and output channel co
//     dst[dpix(p)][co] = epi( bias[co] + sum_t sum_ci pro(src[spix(p,t)][ci]) * Wp[t][ci][co] )
// with NHWC fp32 activations, K = (taps x source channels) contracted on the matrix cores.
// Weight gradients are the transposed contraction over pixels (lf_tapwgrad).
#pragma once
#include "lf_common.h"

#define LF_MAX_TAPS 9

struct LfTapGeom {
    int N, Hl, Wl;                 // logical pixel grid the kernel iterates over
    int Hs, Ws, s_pix, s_choff;    // source tensor: spatial dims, floats per pixel, channel offset
    int ssh, ssw;                  // source coord = logical * ss + tap offset
    int Hd, Wd, d_pix, d_choff;    // destination tensor
    int dsh, dsw, dah, daw;        // dest coord = logical * ds + da
    int Cs, Cd;                    // contracted channels (multiple of 16), produced channels (multiple of 16)
    int ntaps;
    int tdh[LF_MAX_TAPS], tdw[LF_MAX_TAPS];
};

enum { LF_PRO_NONE = 0, LF_PRO_BNRELU = 1 };
enum {
    LF_EPI_RELU = 1,        // v = max(v, 0)
    LF_EPI_MASK = 2,        // v = mask_src > 0 ? v : 0        (ReLU backward from the saved output)
    LF_EPI_ADD = 4,         // v += add_src                    (residual / accumulated gradient)
    LF_EPI_STATS_SQ = 8,    // per-channel sum v, sum v^2      (BatchNorm forward statistics)
    LF_EPI_MASKBN = 16,     // v = (aux*msc+msh) > 0 ? v : 0   (ReLU backward through a recomputed BN)
    LF_EPI_STATS_XHAT = 32  // per-channel sum v, sum v*aux (RAW: lf_bn_bwd_finalize turns the pair into sum v*xhat, xhat = aux*rstd - mean*rstd, in fp64)
};

struct LfTapArgs {
    const float* src;
    const float* wp;        // packed weights [tap][Cs/4][Cd][4]
    int s16;                // 1: src / dst / mask_src / add_src / aux hold bf16 elements (needs wp16)
    const void* wp16;       // non-null selects the bf16 matrix-core kernel: packed bf16 weights [tap][ceil(Cs/32)*4][Cd][8]
    int split;              // 9: fp32 on the bf16 matrix cores from 3-way split operands, all nine partial products (tapgemm_split_kernel);
    const void* wp48;       //   its weights, split by the pack kernel: bf16 [tap][Cs/8][Cd][3][8]; launches the split
                            //   kernel cannot take (Cs % 32, Cd % 64) fall back to the fp32 matrix cores (wp)
    const float* bias;      // [Cd] or null
    float* dst;
    const float* pro_sc;    // prologue BN scale / shift per source channel
    const float* pro_sh;
    const float* mask_src;  // dest-geometry tensors
    const float* add_src;
    const float* aux;
    const float* msc;       // [Cd] forward BN scale/shift (MASKBN)
    const float* msh;
    const float* asc;       // (unused since round 3: the BatchNorm-backward sums are written raw)
    const float* ash;
    const float* dm;        // optional Dropout2d keep-mask [N][Cd] applied to the STATS_XHAT sums only (gm = v * dm)
    float* stats;           // per-workgroup partial sums, channel-major: [2][Cd][stats_ld], row r = 256-pixel tile r of the launch
    int stats_ld;           //   (rows = lf_tapgemm_stat_rows(); stats_ld >= rows: lf_eltwise.h, LfStatPart)
    unsigned long long* dbg;  // optional: per-wave phase timestamps (s_memtime), 8 words per wave (tools/kbench.py --phases)
};

void lf_tapgemm_set_split_any_size(int v);
void lf_tapgemm_set_bf16_lds(int v);
int lf_tapgemm_stat_rows(const LfTapGeom& g);                          // upper bound over the kernels (buffer sizing)
int lf_tapgemm_stat_rows_for(const LfTapGeom& g, const LfTapArgs& a);  // rows the launch with these arguments writes
int lf_tapgemm_launch(const LfTapGeom& g, const LfTapArgs& a, int pro, int epi, hipStream_t st);
// the same launch WITHOUT the in-order barrier on its dispatch packet (hipExtAnyOrderLaunch; fp32 tap-GEMM kernels only, the other
// kernels launch in order): it may overlap the previous launch of the stream, which the caller guarantees to be independent of it
int lf_tapgemm_launch_unordered(const LfTapGeom& g, const LfTapArgs& a, int pro, int epi, hipStream_t st);

struct LfWgradArgs {
    const float* x;         // source-side tensor (gathered by taps), geometry = source fields of LfTapGeom
    const float* g;         // dest-side gradient tensor, geometry = dest fields
    const float* pro_sc;    // optional BN+ReLU recompute on x
    const float* pro_sh;
    int s16;                // 1: x and g hold bf16 elements (partials and bias rows stay fp32)
    int split;              // (unused since round 6: the weight gradient of the split mode runs on the fp32 matrix cores)
    float* partial;         // [splits][ntaps][Cs][Cd]
    float* bias_partial;    // [bias_rows][Cd] or null
    unsigned long long* dbg = nullptr;   // tools/kbench.py --phases: 8 words per wave (start, first operands, loop done, end, HW id)
};
// number of k-split rows the kernel will write for this geometry
int lf_tapwgrad_splits(const LfTapGeom& g);          // upper bound over the kernels that take fp32 tensors
int lf_tapwgrad_splits_bound(const LfTapGeom& g, int s16);   // ... by storage type: bf16 tensors add the read-once kernel's rows
// rows the launch with these arguments writes (bf16 tensors: the read-once kernel's row count)
int lf_tapwgrad_splits_for(const LfTapGeom& g, const LfWgradArgs& a, int pro);
int lf_tapwgrad_launch(const LfTapGeom& g, const LfWgradArgs& a, int pro, hipStream_t st);

// Read-once weight gradient of the 3-tap C -> C convolutions on bf16 tensors (lf_wgrad_ro.hip; C = 64, 128, stride 1, Wl % 16 == 0):
// one workgroup owns all taps of a 64-channel x-block against every g-channel for its pixel range.  lf_tapwgrad_launch routes to it.
bool lf_tapwgrad_ro_ok(const LfTapGeom& g, int s16);
int lf_tapwgrad_ro_rows_bound(const LfTapGeom& g);     // partial rows at the shipped workgroup caps (buffer sizing; 0 = geometry not taken)
int lf_tapwgrad_ro_rows(const LfTapGeom& g);           // rows the launch writes ([rows][3][C][C] fp32 + [rows][C] bias rows)
int lf_tapwgrad_ro_launch(const LfTapGeom& g, const LfWgradArgs& a, int pro, hipStream_t st);
void lf_tapwgrad_ro_set(int mode, int cap64, int cap128);   // tools / A-B runs: 0 = off; workgroups per launch at 64 / 128 channels

// dst[k*sk + n*sn + tapidx[t]] = sum_s partial[s][t][k][n]
// ... and, in the same launch, bias_grad[n] (+)= sum_r bias_rows[r][n] when bias_rows != null
int lf_wgrad_reduce_launch(const float* partial, int splits, int ntaps, int Cs, int Cd, float* grad, long sk, long sn,
                           const int* tapidx_host, const float* bias_rows, int n_bias_rows, float* bias_grad,
                           int bias_accumulate, hipStream_t st);
// The same reduction for up to LF_REDUCE_BATCH independent weight gradients in ONE launch (the jobs travel as kernel
// arguments): lf_erfnet collects the reductions of a whole backward pass -- each weight gradient keeps its partial rows
// in its own region of the workspace -- and issues them at the end instead of one 6 us launch behind every wgrad.
#define LF_REDUCE_BATCH 32
struct LfReduceJob {
    const float* partial; float* grad; const float* bias_rows; float* bias_grad;
    long sk, sn;
    int splits, ntaps, Cs, Cd, n_bias_rows;
    int tapidx[LF_MAX_TAPS];
};
int lf_wgrad_reduce_batch_launch(const LfReduceJob* jobs_host, int njobs, hipStream_t st);
// dst[n] (+)= sum_r rows[r][n]
int lf_rows_reduce_launch(const float* rows, int nrows, int C, float* dst, int accumulate, hipStream_t st);

// wp[((t*(Kc/4) + k/4)*Nc + n)*4 + k%4] = w[k*sk + n*sn + tapidx[t]]
struct LfPackEntry {
    long src_off;   // float offset into the flat parameter arena / or pointer index, see lf_erfnet
    long dst_off;   // float offset into the packed-weight arena
    int Kc, Nc, ntaps;
    long sk, sn;
    int tapidx[LF_MAX_TAPS];
    int param;      // index of the parameter tensor holding the weights
    long dst16_off; // bf16-element offset into the bf16 packed arena (precision mode "bf16")
};
int lf_pack_weights_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, float* arena,
                           hipStream_t st);
long lf_pack_bf16_elems(int Kc, int Nc, int ntaps);
// split weights: 3 bf16 pieces per element, entry e at 3 * e.dst16_off of arena48 (entries with Kc % 32 != 0 are skipped)
int lf_pack_weights_split_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, void* arena48,
                                 hipStream_t st);
bool lf_tapgemm_split_ok(const LfTapGeom& g);
int lf_pack_weights_bf16_launch(const LfPackEntry* entries_dev, int nentries, const float* const* params_dev, void* arena16,
                                hipStream_t st);
