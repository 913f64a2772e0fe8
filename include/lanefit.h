/*
 * lanefit.h -- C ABI of liblanefit_hip.so, the MI355X (gfx950) implementation of the
 * LaneDetection_End2End hot path: ERFNet backbone -> weighted-least-squares lane fit ->
 * area / back-projection / segmentation losses, forward and backward.
 *
 * The reference has no FFI boundary of its own (it is pure PyTorch): its
 * operator API is the nn.Module surface.  Each entry point below therefore names the
 * reference nn.Module / function it replaces (file:line under /root/reference, with
 * BEV/ = Birds_Eye_View_Loss/, BP/ = Backprojection_Loss/).  The Python host layer in
 * lanedetection_end2end_amd/ binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - plain C types only: no torch / ATen / HIP types in any signature;
 *   - return value: 0 = launches enqueued, <0 = argument error (lf_last_error() has text).
 *     Numerical failure (singular normal matrix) is reported asynchronously through a
 *     device-side `status` word the caller reads back (1 = singular, 2 = not positive
 *     definite) so that the host can raise RuntimeError like torch.inverse does
 *     (BEV/main.py:213-219 catches exactly that);
 *   - activation tensors inside the backbone are NHWC fp32; the module boundary
 *     (input image, logits, weight maps) is NCHW fp32 as in the reference.
 */
#ifndef LANEFIT_H
#define LANEFIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_ABI_VERSION 5      /* 5 (round 6): lf_erfnet_workspace_bytes_for added; lf_erfnet_set_precision refuses the removed modes 1 and 4;
                                 lf_erfnet_forward_range / _backward_range take precision mode 2 */

/* activation applied to the backbone logits: BEV/Networks/LSQ_layer.py:43-63 */
enum { LF_ACT_SQUARE = 0, LF_ACT_ABS = 1, LF_ACT_RELU = 2, LF_ACT_SIGMOID = 3,
       LF_ACT_SOFTPLUS = 4, LF_ACT_NONE = 5 };
/* normal-equation solver: torch.inverse path (LSQ_layer.py:128-130) or GELS/Cholesky (BP/Networks/gels.py) */
enum { LF_SOLVE_LU = 0, LF_SOLVE_CHOLESKY = 1 };
/* Area_Loss weightings: BEV/Loss_crit.py:103-121 */
enum { LF_WF_NONE = 0, LF_WF_LINEAR = 1, LF_WF_QUADRATIC = 2 };
enum { LF_F32 = 0, LF_F64 = 1 };

int lf_abi_version(void);
const char* lf_last_error(void);

/* ------------------------------------------------------------------------------------
 * Fitting head.  Replaces Net.forward steps 2-4 (activation, row mask, grid, WLS):
 * BEV/Networks/LSQ_layer.py:310-325 and Weighted_least_squares.forward :103-167
 * (BP/Networks/LSQ_layer.py:85-154), plus the GELS option (BP/Networks/gels.py:10-15).
 *
 *   logits   (N,K,H,W) fp32 NCHW backbone output `output`
 *   grid_xy  (H*W,2) fp32 (x', y') of every pixel; grid_batch_stride = 0 when shared by
 *            all images (always the case in the reference), else floats between images
 *   zero_rows  rows [0,zero_rows) of every map are masked to 0 (LSQ_layer.py:257-258,316)
 *   order 0..3, reg = --reg_ls added to the diagonal for BOTH solvers (BEV adds it before either factorisation,
 *     LSQ_layer.py:120-126; BP's GELS has no regulariser, gels.py:10-15: its caller passes reg = 0),
 *     y_offset = 1 (BEV :109) or 255 (BP :94)
 *   beta     out (N,K,order+1) fp64, coefficients highest power first
 *   zinv     out (N,K,(order+1)^2) fp64, (Y0^T Y0 + reg I)^-1 -- saved for backward
 *   masked   out (N,K,H,W) fp32 weight maps after activation+mask, or NULL to skip
 *   partials workspace, >= lf_wls_workspace_bytes(N,K,order) bytes
 *   status   out (N*K) int32: 0 ok, 1 singular, 2 not positive definite
 * ---------------------------------------------------------------------------------- */
size_t lf_wls_workspace_bytes(int N, int K, int order);
int lf_wls_fwd(const float* logits, const float* grid_xy, long grid_batch_stride,
               int N, int K, int H, int W, int zero_rows, int order, double reg,
               double y_offset, int act_kind, int solver,
               double* beta, double* zinv, float* masked, void* partials, int32_t* status,
               void* stream);
/* d loss / d logits from d loss / d beta (autograd of the bmm/inverse chain; gels.py:17-25).
 *   grad_beta (N,K,order+1) fp64; lanes whose beta received no gradient pass zeros.
 *   grad_logits out (N,K,H,W) fp32 (masked rows are written as 0). */
int lf_wls_bwd(const float* logits, const float* grid_xy, long grid_batch_stride,
               int N, int K, int H, int W, int zero_rows, int order, double y_offset,
               int act_kind, const double* beta, const double* zinv, const double* grad_beta,
               float* grad_logits, void* stream);

/* GELS.forward / GELS.backward -- BP/Networks/gels.py:9-25: x = (A^T A)^-1 A^T b by Cholesky of the normal
 * equations (no regulariser) and the hand-written backward of the reference.
 *   A (N,P,D) fp32, b (N,P) fp32, D <= 4; x out (N,D) fp32; zinv out (N,D,D) fp64 (saved for backward);
 *   partials >= lf_gels_workspace_bytes(N,D); status out (N) int32 (2 = not positive definite).
 *   backward: grad_out (N,D) fp32 -> grad_A (N,P,D), grad_b (N,P) fp32. */
size_t lf_gels_workspace_bytes(int N, int D);
int lf_gels_fwd(const float* A, const float* b, int N, long P, int D, float* x, double* zinv, void* partials,
                int32_t* status, void* stream);
int lf_gels_bwd(const float* A, const float* b, const float* x, const double* zinv, const float* grad_out,
                int N, long P, int D, float* grad_A, float* grad_b, void* stream);

/* Area_Loss.forward -- BEV/Loss_crit.py:98-134.  beta (N,order+1) with element stride
 * beta_stride between images, gt (N,order+1) contiguous; dtype LF_F32/LF_F64 for both.
 * loss out: 1 element of dtype; grad out: (N,order+1) of dtype = d loss / d beta. */
int lf_area_loss(const void* beta, long beta_stride, const void* gt, int N, int order,
                 int weight_funct, int dtype, void* loss, void* grad, void* stream);

/* MSE_Loss.forward -- BEV/Loss_crit.py:137-150 (BP/Loss_crit.py:147-160): nn.MSELoss() of params.squeeze(-1) against
 * gt_params = mean over ALL n = N * (order + 1) elements of the squared difference (--loss_policy mse).  params, gt: n
 * contiguous elements of dtype (LF_F32 / LF_F64); loss out: 1 element; grad out: n elements = 2 (params - gt) / n. */
int lf_mse_loss(const void* params, const void* gt, long n, int dtype, void* loss, void* grad, void* stream);

/* backprojection_loss.forward -- BP/Loss_crit.py:202-218 (constants of :166-200 passed in).
 *   beta (N,order+1) fp64 (stride beta_stride), x_gt/valid (N,S) fp64, Y (S,order+1) fp64,
 *   y_prime (S) fp64, minv_host: 9 doubles (row-major M^-1) read on the HOST at call time.
 *   loss out fp64 scalar; x_cal_valid out (N,S) fp64; grad out (N,order+1) fp64. */
int lf_backproj_loss(const double* beta, long beta_stride, const double* x_gt, const double* valid,
                     const double* Y, const double* y_prime, const double* minv_host,
                     int N, int S, int order, double* loss, double* x_cal_valid, double* grad,
                     void* stream);

/* Class-weighted pixel cross entropy, weighted-mean reduction: BEV/Loss_crit.py:61-75,
 * BP/Loss_crit.py:64-65.  logits (N,C,H,W) fp32, target (N,H,W) int64, weights (C) fp32.
 * acc: 3 doubles of scratch: weighted loss sum, weight sum, and the NUMBER OF TARGETS OUTSIDE [0, C) -- those pixels get
 * weight 0 in loss and gradient, and the host raises when the count is non-zero (torch's NLLLoss asserts on them).
 * loss out fp32 scalar. */
int lf_ce2d_fwd(const float* logits, const int64_t* target, const float* weights,
                int N, int C, int H, int W, double* acc, float* loss, void* stream);
/* grad_logits (N,C,H,W) = upstream * d loss / d logits; `acc` as left by lf_ce2d_fwd. */
int lf_ce2d_bwd(const float* logits, const int64_t* target, const float* weights,
                int N, int C, int H, int W, const double* acc, const float* upstream,
                float* grad_logits, void* stream);

/* ------------------------------------------------------------------------------------
 * ERFNet backbone engine.  Replaces ERFNet.Net.forward (BEV/Networks/ERFNet.py:151-157;
 * DownsamplerBlock :11-22, non_bottleneck_1d :25-60, Encoder :63-95, UpsamplerBlock :98-107,
 * Decoder :109-142; BP/Networks/ERFNet.py is the same network) and its autograd backward.
 *
 * A plan is built once per input shape (host-side object, no device allocation); every call
 * runs the whole pass on `stream` out of a caller-owned workspace that must stay untouched
 * between a forward and its backward (it holds the saved activations, NHWC fp32).
 *
 *   params: the model's parameter tensors in state_dict() order, buffers excluded (weights,
 *     biases, BN weight/bias; 228 of them for one head, +2 with the `pretrained` second head),
 *     given both as a HOST array of device pointers and as a DEVICE array of the same pointers;
 *   running: 2 * lf_erfnet_num_bn() device pointers (running_mean, running_var per BN, module
 *     order); updated in training mode exactly like nn.BatchNorm2d (momentum 0.1, eps 1e-3);
 *   dropmask: Dropout2d keep-masks (0 or 1/(1-p) per (image, channel)) for the 13 encoder blocks
 *     with p > 0 (ERFNet.py:41,57-58), laid out per lf_erfnet_dropmask_offset(); NULL = no dropout;
 *   head: 0 = decoder.output_conv, 1 = decoder.output_conv2 (Decoder.forward flag, :134-141);
 *   logits: (N, out_channels + head, H, W) fp32 NCHW.
 * ---------------------------------------------------------------------------------- */
typedef struct lf_erfnet_plan lf_erfnet_plan;
lf_erfnet_plan* lf_erfnet_plan_create(int N, int H, int W, int in_channels, int out_channels, int n_heads);
void lf_erfnet_plan_destroy(lf_erfnet_plan* plan);
/* Precision mode of the backbone (there is no reference for modes 2 and 3 -- the reference is fp32 only):
 * 0 = fp32 matrix cores (v_mfma_f32_16x16x4_f32), fp32 tensors: the default, the parity path, the BASELINE headline;
 * 2 = bf16 tensors (BASELINE config 3): every activation / gradient tensor of the workspace stored as bf16, convolutions, data
 *     gradients and weight gradients on the bf16 matrix cores with fp32 accumulation; parameters, their gradients, BatchNorm
 *     statistics and the logits stay fp32;
 * 3 = fp32 results on the bf16 matrix cores: tensors and accumulation as in mode 0, every product of the 64- and 128-channel
 *     convolutions and data gradients formed from exact 3-way bf16 splits of both fp32 operands (all 9 partial products, i.e.
 *     exact products); parity as mode 0 (tests/test_backbone_gpu.py); launches the split kernel cannot take (other channel
 *     counts, pixel counts not a multiple of 512) and the weight gradient run on the fp32 matrix cores.
 * (Modes 1 and 4 -- bf16 operands with fp32 tensors, and the 6-term split -- existed through ABI 4; no BASELINE configuration
 * used them and they were removed in round 6: the call returns an error for them.) */
int lf_erfnet_set_precision(const lf_erfnet_plan* plan, int mode);
size_t lf_erfnet_workspace_bytes(const lf_erfnet_plan* plan);   /* follows the precision mode (bf16 tensors: larger partial-row regions): query it after lf_erfnet_set_precision */
size_t lf_erfnet_workspace_bytes_for(const lf_erfnet_plan* plan, int mode);   /* the same for a named mode, whatever the plan's current setting (0: unknown mode) */
long lf_erfnet_activation_floats(const lf_erfnet_plan* plan);   /* elements of the saved activations (all layers): bench.py's HBM roofline */
int lf_erfnet_num_params(const lf_erfnet_plan* plan);
int lf_erfnet_num_bn(const lf_erfnet_plan* plan);
int lf_erfnet_num_dropout(const lf_erfnet_plan* plan);
long lf_erfnet_dropmask_floats(const lf_erfnet_plan* plan);
long lf_erfnet_dropmask_offset(const lf_erfnet_plan* plan, int i);      /* float offset of block i */
int lf_erfnet_dropmask_channels(const lf_erfnet_plan* plan, int i);
long lf_erfnet_encoder_offset(const lf_erfnet_plan* plan);             /* float offset of the encoder output (N,H/8,W/8,128) NHWC */
long lf_erfnet_activation_offset(const lf_erfnet_plan* plan, int layer, int slot);
/* float offset of a BatchNorm's folded per-channel fp32 vector in the workspace (parity tests: the values the forward pass decided
 * its ReLU masks with).  bn: 0 / 1 within the layer (non_bottleneck_1d: bn1, bn2); which: 0 scale = gamma * rstd, 1 shift =
 * beta - mean * scale, 2 rstd, 3 -mean * rstd.  Valid after lf_erfnet_forward; -1 when out of range. */
long lf_erfnet_bn_vector_offset(const lf_erfnet_plan* plan, int layer, int bn, int which);
/* head = 0 / 1 selects output_conv / output_conv2; head = -1 = ENCODER ONLY (Net.forward(only_encode=True),
 * BEV/Networks/ERFNet.py:151-153): the decoder does not run (its BatchNorm running statistics stay untouched, as in the
 * reference, where the decoder is never called on that branch), logits may be NULL; the matching backward takes
 * head = -1 too, with grad_encoder as the incoming gradient. */
int lf_erfnet_forward(const lf_erfnet_plan* plan, const float* img, const float* const* params_host,
                      const float* const* params_dev, float* const* running_host, const float* dropmask,
                      int training, int head, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* Gradients are written (not accumulated) into grads_host[i]; NULL entries are skipped.
 * grad_encoder: optional (N,H/8,W/8,128) NHWC gradient w.r.t. the encoder output (`shared_encoder`,
 * BP/Networks/LSQ_layer.py:275), added where the decoder's gradient reaches the encoder; NULL = none.
 * training: the mode the matching forward ran in.  1 = batch statistics (autograd of nn.BatchNorm2d in train mode);
 * 0 = running statistics (net.eval() with gradients enabled, e.g. fine-tuning with frozen statistics): BatchNorm is then
 * a per-channel affine map and its data gradient is gamma * rstd * dy, as torch computes it. */
int lf_erfnet_backward(const lf_erfnet_plan* plan, const float* img, const float* grad_logits,
                       const float* grad_encoder, const float* const* params_host, float* const* grads_host,
                       const float* dropmask, int training, int head, void* workspace, size_t workspace_bytes, void* stream);
int lf_nhwc_to_nchw(const float* src, float* dst, int N, int H, int W, int C, void* stream);

/* Block-level surface (round 4): a contiguous range [first, last) of the plan's layers (module order: 0 =
 * encoder.initial_block, 1..15 = encoder.layers[0..14], 16..21 = decoder.layers[0..5]) as one call, inside the plan of the
 * whole network at the matching input size -- what makes the reference's sub-modules callable on their own:
 *   DownsamplerBlock.forward(input)            BEV/Networks/ERFNet.py:19-22
 *   non_bottleneck_1d.forward(input)           :44-60
 *   UpsamplerBlock.forward(input)              :104-107
 *   Encoder.forward(input, predict=False)      :86-95   (range 0..16; predict = lf_pointwise_fwd on top)
 *   Decoder.forward(input, flag)               :129-142 (range 16..22 with head = 0 / 1)
 * x / y / gy / gx are NCHW fp32 like the reference's tensors; head >= 0 (only with last = the layer count) appends
 * output_conv (0) / output_conv2 (1) and makes y the logits.  Only the range's parameters are read / receive gradients
 * (grads_host entries outside it are left untouched); BatchNorm running statistics of the range are updated in train mode.
 * lf_erfnet_backward_range must follow lf_erfnet_forward_range on the same workspace.  Every precision mode (in mode 2 the
 * range's input and the incoming gradient are rounded to bf16 on their way into the workspace, outputs widened on their way out).
 * The workspace of a range is COMPACT (ABI 4): lf_erfnet_range_workspace_bytes(plan, first, last) = the range's own activations
 * + the plan's globals (packed weights, statistics rows, one partial-row region, three gradient buffers), so that a loop over
 * the blocks of a network keeps the activations of ONE network alive until backward, not one whole-network workspace per block.
 * gx is not produced for first = 0: the image takes no gradient (as in lf_erfnet_backward). */
int lf_erfnet_num_layers(const lf_erfnet_plan* plan);
size_t lf_erfnet_range_workspace_bytes(const lf_erfnet_plan* plan, int first, int last);
int lf_erfnet_layer_io(const lf_erfnet_plan* plan, int layer, int* out6_host);   /* Cin, Hin, Win, Cout, Hout, Wout */
int lf_erfnet_forward_range(const lf_erfnet_plan* plan, int first, int last, int head, const float* x,
                            const float* const* params_host, const float* const* params_dev, float* const* running_host,
                            const float* dropmask, int training, float* y, void* workspace, size_t workspace_bytes, void* stream);
int lf_erfnet_backward_range(const lf_erfnet_plan* plan, int first, int last, int head, const float* x, const float* gy,
                             const float* const* params_host, float* const* grads_host, const float* dropmask, int training,
                             float* gx, void* workspace, size_t workspace_bytes, void* stream);
/* encoder.output_conv = nn.Conv2d(128, num_classes, 1): the `predict=True` branch of Encoder.forward that
 * Net.forward(input, flag, only_encode=True) returns (BEV/Networks/ERFNet.py:84,86-95,151-153).
 * x (N,h,w,C) NHWC fp32 (the encoder output, read in place), w (K,C) = the weight (K,C,1,1), b (K) or NULL -> y (N,K,h,w) NCHW.
 * Backward: gy (N,K,h,w) -> gx (N,h,w,C) NHWC (or NULL), gw (K,C) and gb (K) (or NULL); scratch >= lf_pointwise_scratch_floats(). */
int lf_pointwise_fwd(const float* x, const float* w, const float* b, float* y, int N, int h, int w_, int C, int K, void* stream);
long lf_pointwise_scratch_floats(int N, int h, int w_, int C, int K);
int lf_pointwise_bwd(const float* x, const float* gy, const float* w, float* gx, float* gw, float* gb, int N, int h, int w_,
                     int C, int K, float* scratch, void* stream);
/* ------------------------------------------------------------------------------------
 * Cold-path arithmetic natively (round 4): the nn.Linear tails of the --clas heads and the segmentation-mode fit input.
 *
 * lf_linear_fwd: y (N,O) = act(x (N,K) @ w (O,K)^T + b (O) or NULL), act = identity or ReLU (relu = 1).  Replaces
 *   F.relu(self.fully_connected1(x)), self.fully_connected_line1(x), self.fully_connected_horizon(x)
 *   (BP/Networks/LSQ_layer.py:186-189,199-207; BEV/Networks/LSQ_layer.py:198-205,218-226).  fp32, fixed summation order.
 * lf_linear_bwd: gy (N,O) [+ y, the forward's output, when relu] -> gx (N,K), gw (O,K), gb (O); NULL outputs are skipped
 *   (gb needs gw).  Written, not accumulated.
 * lf_seg_maps: logits (N,C,H,W) -> maps (N,L,H,W): per-lane maps valued k where the arg-max over the C class logits is k
 *   (first maximum, like torch.max), rows < zero_rows zeroed (index_fill), lanes flagged in gt_line (N,L fp32, or NULL) set to
 *   map [0,0] -- Net.forward(end_to_end=False): BEV/Networks/LSQ_layer.py:302-308,316; BP/Networks/LSQ_layer.py:279-293,298,308-311. */
int lf_linear_fwd(const float* x, const float* w, const float* b, float* y, int N, int K, int O, int relu, void* stream);
int lf_linear_bwd(const float* x, const float* w, const float* y, const float* gy, float* gx, float* gw, float* gb, int N, int K,
                  int O, int relu, void* stream);
int lf_seg_maps(const float* logits, const float* gt_line, float* maps, int N, int C, int L, int H, int W, int zero_rows,
                void* stream);
/* Roofline instrumentation (bench.py): HIP event pairs around every matrix-core launch of the engine.
 * out6 = {ms, algorithmic FLOPs, launches} for family 0 (tap-GEMM forward + data gradient) and
 * family 1 (weight gradient), accumulated since the last read. */
int lf_erfnet_profile(const lf_erfnet_plan* plan, int enable);
/* csv_path_host: optional file receiving one line per recorded launch (family, layer, shape, us, TFLOP/s); NULL = none */
int lf_erfnet_profile_read(const lf_erfnet_plan* plan, double* out6_host, const char* csv_path_host);

/* ------------------------------------------------------------------------------------
 * Fused Adam step over all parameter tensors in ONE launch ("next" row 8f-1).  Replaces optimizer.step() of the
 * torch.optim.Adam the reference builds (BEV/Networks/utils.py:411-420, BEV/main.py:266); same arithmetic.
 *   tensors_dev: n records {float* p; const float* g; float* m; float* v; long numel; long step[2]} on the device: the
 *     step count is PER TENSOR like torch.optim.Adam's (a head that starts training late has its own bias correction) and
 *     lives in the record, double-buffered;
 *   work_dev: nblocks int2 {tensor index, chunk index}, chunk = lf_adam_chunk() elements;
 *   parity: the slot of step[] holding each tensor's count of updates so far; the launch writes count + 1 to the other slot
 *     (alternate parity between calls); grad_scale: multiplies every gradient first (1/world size).
 * ---------------------------------------------------------------------------------- */
int lf_adam_chunk(void);
int lf_adam_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int parity, float grad_scale, void* stream);
/* The other optimizers of define_optim (BEV/Networks/utils.py:414-417), same tables (record field m = momentum buffer,
 * v = RMSprop's square average; the step slots are unused): torch.optim.SGD(momentum, dampening 0, no Nesterov) and
 * torch.optim.RMSprop(alpha, eps, momentum, not centered), L2 weight decay folded into the gradient as torch does. */
int lf_sgd_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float momentum, float weight_decay,
                float grad_scale, void* stream);
int lf_rmsprop_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float alpha, float eps,
                    float momentum, float weight_decay, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------
 * "Next" row 8f-3: the `--clas` heads and the inference-side post-processing of BP/test.py.
 *
 * lf_convchain_*: a chain of [Conv2d(k, stride 1, pad k/2, bias) -> BatchNorm2d -> ReLU] blocks on an NHWC fp32
 * tensor = the trunk of `Classification` (BP/Networks/LSQ_layer.py:150-181,192-196: conv1..conv4 + BNs) that the
 * line-type and horizon heads run on the encoder output.  Plan per shape; channels[0..L] multiples of 16,
 * ksize[i] in {1,3}.  params: 4 per block (conv.weight (Co,Ci,k,k), conv.bias, bn.weight, bn.bias) as a HOST
 * and a DEVICE array of device pointers; running_host: 2 per block.  x is read in place from the backbone
 * workspace (lf_erfnet_encoder_offset); y = last block's post-ReLU output, NHWC.  The workspace holds the
 * saved pre-BN tensors until lf_convchain_backward, which writes every parameter gradient and (if gx != NULL)
 * the NHWC input gradient to hand to lf_erfnet_backward(grad_encoder).
 * ---------------------------------------------------------------------------------- */
typedef struct lf_convchain_plan lf_convchain_plan;
lf_convchain_plan* lf_convchain_plan_create(int N, int H, int W, int nlayers, const int* channels_host,
                                            const int* ksize_host);
void lf_convchain_plan_destroy(lf_convchain_plan* plan);
size_t lf_convchain_workspace_bytes(const lf_convchain_plan* plan);
int lf_convchain_forward(const lf_convchain_plan* plan, const float* x, const float* const* params_host,
                         const float* const* params_dev, float* const* running_host, int training, float momentum,
                         float eps, float* y, void* workspace, size_t workspace_bytes, void* stream);
/* training = the mode of the matching forward (0: running statistics; see lf_erfnet_backward) */
int lf_convchain_backward(const lf_convchain_plan* plan, const float* x, const float* y, const float* gy,
                          const float* const* params_host, float* const* grads_host, float* gx, int training,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Pool + flatten in front of the heads' fully connected layers (LSQ_layer.py:183-187,197-201):
 * mode 0 = MaxPool2d(2,2) -> (N, C*(H/2)*(W/2)); mode 1 = AvgPool2d((1,W)) -> (N, C*H); input NHWC,
 * output in the NCHW flatten order nn.Linear's weights expect.  The Linear layers themselves are plain
 * library GEMMs (rocBLAS through torch.nn.functional.linear). */
int lf_poolflat_fwd(const float* y, int N, int H, int W, int C, int mode, float* out, void* stream);
int lf_poolflat_bwd(const float* y, const float* gout, int N, int H, int W, int C, int mode, float* gy, void* stream);

/* Projections.compute_coordinates (BP/test.py:172-186) + the gating of test_model (:77-88) in one launch:
 * x = resize(M_inv . [poly(beta, y_eval), y_prime, 1]) per (image, lane, sample height); then
 * line_flag[n][l] == 0 -> fill; sample index < bound[n] (python slice semantics) -> fill; x > hi or x < lo -> fill.
 * beta (N,L,order+1) fp64, highest power first; y_eval / y_prime (S) fp64; minv_host 9 doubles;
 * line_flag (N,L) fp32 or NULL; bound (N) int32 or NULL; lo > hi disables the range gate.
 * Outputs (either may be NULL): x_out (N,L,S) fp64, x_int (N,L,S) int32 = round-half-even(x). */
int lf_lane_decode(const double* beta, const double* y_eval, const double* y_prime, const double* minv_host,
                   double scale, const float* line_flag, const int* bound, double lo, double hi, double fill,
                   int N, int L, int S, int order, double* x_out, int* x_int, void* stream);

/* "Next" row 8f-2: polynomial.trapezoidal (BEV/Loss_crit.py:26-35): area between two parabolas by the
 * trapezium rule on [a, b] with n intervals; p, q (B,3) rows [a2, a1, a0]; fp32 or fp64; out (B). */
int lf_trapezoid(const void* p, const void* q, int B, double a, double b, int n, int is_double, void* out,
                 void* stream);

/* ------------------------------------------------------------------------------------
 * "Next" row 8f-4: the input pipeline on the device.  Replaces the per-sample PIL/torchvision work of
 * LaneDataset.__getitem__ (BEV/Dataloader/Load_Data_new.py:77-101, BP/Dataloader/Load_Data_new.py:126-173):
 * F.crop(bottom rows) -> F.resize((R,2R), BILINEAR | NEAREST) -> class remap -> F.hflip -> ToTensor, with
 * Pillow's fixed-point resampling arithmetic reproduced bit for bit (see csrc/lf_pipeline.hip).
 *   plan: host object holding the resampling tables for one geometry; upload them once to a device buffer of
 *     lf_pipeline_table_bytes() and pass that buffer to every call;
 *   lf_pipeline_image: frames (N,Hin,Win,3) uint8 HWC -> out (N,3,out_h,out_w) fp32 in [0,1];
 *   lf_pipeline_label: labels (N,Hin,Win) uint8 -> out (N,1,out_h,out_w) int64 through lut (256 int64 =
 *     (ToTensor(v)*255).long()), mode bit 0: zero classes 3,4 (BEV; BP with nclasses < 3), bit 1: BP's flip
 *     statement order (pre-flip masks of 3/4 applied to the flipped map, Load_Data_new.py:152-165);
 *     horizon (N,out_h) fp32 = ones above the first labelled row (BEV :103-105) or NULL;
 *   flip: (N) uint8 per-sample horizontal flip flags or NULL.
 * ---------------------------------------------------------------------------------- */
typedef struct lf_pipeline_plan lf_pipeline_plan;
lf_pipeline_plan* lf_pipeline_plan_create(int Hin, int Win, int crop_top, int crop_h, int out_h, int out_w);
void lf_pipeline_plan_destroy(lf_pipeline_plan* plan);
size_t lf_pipeline_table_bytes(const lf_pipeline_plan* plan);
int lf_pipeline_upload(const lf_pipeline_plan* plan, void* tables_dev, void* stream);
int lf_pipeline_tables_host(const lf_pipeline_plan* plan, int* ksx_host, int* ksy_host, int* bx_host, int* kx_host,
                            int* by_host, int* ky_host, int* ntx_host, int* nty_host);
int lf_pipeline_image(const lf_pipeline_plan* plan, const uint8_t* frames, int N, const void* tables_dev,
                      const uint8_t* flip, float* out, void* stream);
int lf_pipeline_label(const lf_pipeline_plan* plan, const uint8_t* labels, int N, const void* tables_dev,
                      const uint8_t* flip, int mode, const int64_t* lut, int64_t* out, float* horizon, void* stream);
/* The same two calls reading a RESIDENT POOL of decoded frames / label maps (pool_frames entries): batch element n is pool entry
 * sel[n] (int64 on the device).  The index batch of a cached dataset (SubsetRandomSampler's draw,
 * BEV/Dataloader/Load_Data_new.py:305-320) is gathered inside the kernels instead of by a copy of N frames first.
 * bad_index (device int32, may be NULL): incremented once per batch element whose sel[n] lies outside [0, pool_frames); such an
 * element reads pool entry 0 -- never out of bounds -- and the host raises from the count, as the reference's tensor indexing
 * raises IndexError (ABI 4: pool_frames on the label call, bad_index on both). */
int lf_pipeline_image_indexed(const lf_pipeline_plan* plan, const uint8_t* pool, long pool_frames, const int64_t* sel, int N,
                              const void* tables_dev, const uint8_t* flip, float* out, int* bad_index, void* stream);
int lf_pipeline_label_indexed(const lf_pipeline_plan* plan, const uint8_t* pool, long pool_frames, const int64_t* sel, int N,
                              const void* tables_dev, const uint8_t* flip, int mode, const int64_t* lut, int64_t* out, float* horizon,
                              int* bad_index, void* stream);

/* ------------------------------------------------------------------------------------
 * Kernel-level entry points: one factorised convolution of non_bottleneck_1d
 * (nn.Conv2d(C, C, (3,1)|(1,3), padding = dilation = d), BEV/Networks/ERFNet.py:29-37) on NHWC fp32
 * tensors, outside the plan: forward, data gradient (optionally times the ReLU mask of mask_src),
 * weight + bias gradient.  w / gw use the nn.Conv2d layout (C, C, 3) flattened; axis 0 = 3x1 (along H),
 * 1 = 1x3 (along W); C a multiple of 16; scratch >= lf_conv1d_scratch_floats() floats.
 * ---------------------------------------------------------------------------------- */
long lf_conv1d_scratch_floats(int N, int H, int W, int C);
int lf_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C,
                  int axis, int dilation, int relu, float* scratch, void* stream);
int lf_conv1d_bwd_data(const float* gy, const float* w, const float* mask_src, float* gx, int N, int H, int W,
                       int C, int axis, int dilation, float* scratch, void* stream);
int lf_conv1d_bwd_weight(const float* x, const float* gy, float* gw, float* gb, int N, int H, int W, int C,
                         int axis, int dilation, float* scratch, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* LANEFIT_H */
