// Bandwidth-bound kernels of the ERFNet backbone.  Activation / gradient tensors are NHWC fp32, or NHWC bf16 when the
// launcher's `s16` flag is set (precision mode 2; the pointers still travel as float*): arithmetic is fp32 either
// way, per-channel vectors, statistics rows and parameter gradients are always fp32.  Kernels: BatchNorm statistics finalise /
// apply / backward, max-pool branches, the 3-channel stem, the 2x2 transposed-conv head.
#pragma once
#include "lf_common.h"

struct LfStatPart {   // one source of per-channel partial sums, CHANNEL-MAJOR (round 6): [2][C][ld] floats, element (kind, c, row)
                      // = rows[(kind * C + c) * ld + row]; ld >= nrows, a multiple of 4 (the finalise kernels take four rows per load)
    const float* rows; int nrows; int C; int ch_off; int ld;
    // tile_pix = 0: RAW rows [sum v][sum v^2] (or [sum g][sum g t] for the backward sums).
    // tile_pix > 0: CENTRED rows of the tap-GEMM epilogues (round 6): [sum v][M2 = sum (v - mean_row)^2] where row r covers pixels
    // (r % seg_rows) * tile_pix .. + tile_pix - 1 of a launch of seg_pix pixels (the four sub-pixel phases of a transposed
    // convolution lay their rows end to end: seg_rows rows per phase); the finalise kernel merges the rows in fp64,
    // sum v^2 = M2_r + (sum v)_r^2 / n_r, so the fp32 partials never hold a sum of squares about the origin.
    int tile_pix = 0; int seg_rows = 0; long seg_pix = 0;
};
inline int lf_stat_ld(int nrows) { return (nrows + 3) / 4 * 4; }      // the leading dimension the engine gives a use of the statistics buffers
inline LfStatPart lf_stat_part_tiles(const float* rows, int nrows, int ld, int C, int ch_off, int seg_rows, long seg_pix, int tile_pix = 256) {
    LfStatPart p = {rows, nrows, C, ch_off, ld};
    p.tile_pix = tile_pix; p.seg_rows = seg_rows; p.seg_pix = seg_pix;
    return p;
}

// Forward BN: partial rows (raw or centred, see LfStatPart) -> mean/rstd, scale = gamma*rstd, shift = beta - mean*scale;
// train: batch stats + running-stat update (momentum, unbiased var); eval: running stats.
int lf_bn_finalize_fwd(const LfStatPart* parts, int nparts, int C, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, int training,
                       float* scale, float* shift, float* asc /*rstd*/, float* ash /*-mean*rstd*/, hipStream_t st);
// y = relu((x*sc+sh) [* dm[n][c]] [+ res])
int lf_bn_act(const float* x, const float* sc, const float* sh, const float* dm, const float* res, float* y,
              long npix, int C, long pix_per_image, int s16, hipStream_t st);
// gm = g * [y > 0] * dm ; partial rows of (sum gm, sum gm * t) -- RAW second sum, t = the pre-BatchNorm tensor (the finalise
// kernel centres it in fp64: sum gm * xhat = rstd * sum(gm t) - mean * rstd * sum(gm)).  Round 4 measured the centred form
// (sum gm * (t - fl32(mean)), one extra vector and subtraction per element) against it with 50-sigma channel offsets: no
// difference in any parameter gradient -- what degrades there is the FORWARD variance (E[y^2] - mean^2 from fp32 partials),
// DESIGN.md section 2 -- so the raw form, which keeps the epilogues free of per-channel vectors, stays.
int lf_bn_bwd_reduce_rows(long npix);
int lf_bn_bwd_reduce(const float* g, const float* y, const float* t, const float* asc, const float* ash, const float* dm,
                     float* rows, int ld, long npix, int C, long pix_per_image, int s16, hipStream_t st);
// partial rows -> c1 = sum/M, c2 = sumx/M (both 0 when the forward ran in eval mode: running statistics, no mean terms),
// and the parameter gradients ggamma = sumx, gbeta = sum
// rows = [sum g, sum g * t] (RAW, t = the pre-BatchNorm tensor; asc / ash = rstd, -mean * rstd turn them into sum g * xhat in fp64)
int lf_bn_bwd_finalize(const LfStatPart* parts, int nparts, int C, double count, const float* asc, const float* ash, float* c1,
                       float* c2, float* ggamma, float* gbeta, int training, hipStream_t st);
// g_t = gamma*rstd*(gm - c1 - xhat*c2); optionally g_z = g*[y>0]
int lf_bn_bwd_apply(const float* g, const float* y, const float* t, const float* asc, const float* ash, const float* gamma,
                    const float* c1, const float* c2, const float* dm, float* g_t, float* g_z, long npix, int C,
                    long pix_per_image, int s16, hipStream_t st);

// DownsamplerBlock pool branch: cat[..., choff:choff+Cin] = maxpool2x2(x), + stat partial rows
int lf_pool_rows(long npix_out);
int lf_pool_concat_fwd(const float* x, int N, int H, int W, int Cin, float* cat, int cat_pix, int choff, float* rows, int ld,
                       int s16, hipStream_t st);
int lf_pool_bwd(const float* x, const float* gcat, int N, int H, int W, int Cin, int cat_pix, int choff, float* gx,
                int s16, hipStream_t st);

// stem: DownsamplerBlock(3,16) on the NCHW input image: conv3x3 s2 (3->13) || maxpool -> cat (N,H/2,W/2,16)
int lf_stem_rows(int N, int H, int W);
int lf_stem_fwd(const float* img, int N, int Cin, int H, int W, const float* w, const float* b, float* cat, float* rows, int ld,
                int s16, hipStream_t st);
int lf_stem_wgrad_rows(int N, int H, int W);
int lf_stem_wgrad(const float* img, const float* gcat, int N, int Cin, int H, int W, float* wrows, float* brows,
                  int s16, hipStream_t st);

// head: ConvTranspose2d(16, K, 2, stride 2): NHWC (N,h,w,16) -> NCHW (N,K,2h,2w)
int lf_head_fwd(const float* x, const float* w, const float* b, float* out, int N, int h, int w_, int K, int s16,
                hipStream_t st);
int lf_head_bwd_data(const float* gout, const float* w, float* gx, int N, int h, int w_, int K, int s16, hipStream_t st);
int lf_head_wgrad_rows(int N, int h, int w_);
int lf_head_wgrad(const float* x, const float* gout, float* wrows, float* brows, int N, int h, int w_, int K,
                  int s16, hipStream_t st);
