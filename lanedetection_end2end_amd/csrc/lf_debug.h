// Test / tooling hooks of liblanefit_hip.so (NOT part of the public C ABI in include/lanefit.h): the phase-stamp entries of
// tools/kbench.py and the precision selector of the kernel-level lf_conv1d_* parity tests.  Exported with C linkage so
// that ctypes can reach them; process-global state, never touched by the product path.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* 1: the split kernels also take launches below their shipped size rule (kernel-level tests at small shapes) */
void lf_debug_set_split_any_size(int v);
/* which bf16-tensor tap-GEMM kernels run: 4 (shipped) wave-private kernel at 64 channels, whole-line kernel at 128, 16-channel kernels
   where they apply, else the ring; 3 the whole-line kernel at 64 channels too (round 5's routing); 2 the ring for every launch it takes;
   0 the streaming kernel only (A/B timing, bit-identity of the forms: tools/bf16_ab.py, tests/test_bf16_kernels_gpu.py) */
void lf_debug_set_bf16_lds(int v);
/* precision mode of the lf_conv1d_* calls: 0 fp32, 2 bf16 matrix cores on bf16 tensors (x, y, gx, gy, mask_src hold bf16; w, bias,
 * gw, gb stay fp32), 9 fp32 from 9-term split operands (modes 1 and 6 were removed in round 6 and select 0) */
void lf_debug_set_ops_precision(int mode);
/* lf_conv1d_fwd + per-wave s_memrealtime stamps (100 MHz) (start, tap table built, main loop done, stores
 * retired; 8 words per wave) */
int lf_debug_conv1d_fwd_phases(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C,
                               int axis, int dilation, float* scratch, unsigned long long* dbg, void* stream);
/* the weight-gradient launch of lf_conv1d_bwd_weight with the same stamps (start, first operands, main loop done, partials
 * stored, HW id); returns the number of waves launched, -1 on error */
// lf_conv1d_fwd with the BN+ReLU operand prologue and the ReLU epilogue (the third convolution of a non_bottleneck_1d block):
// y = relu(conv1d(relu(x * sc + sh)) + bias); sc, sh: [C] fp32.  Kernel-level tests of the prologue forms (tests/test_bf16_kernels_gpu.py)
int lf_debug_conv1d_fwd_pro(const float* x, const float* w, const float* bias, const float* sc, const float* sh, float* y, int N, int H,
                            int W, int C, int axis, int dilation, float* scratch, void* stream);
/* the read-once bf16 weight gradient (lf_wgrad_ro.hip): mode 0 = off (tapwgrad_kernel's job form takes every launch), 1 = shipped;
 * cap64 / cap128 > 0: workgroups per launch at 64 / 128 channels (A/B runs; at most the shipped 512 / 256 the buffers are sized for) */
void lf_debug_set_wgrad_ro(int mode, int cap64, int cap128);
/* lf_conv1d_bwd_weight with the BN+ReLU operand prologue on x (the weight gradient of a non_bottleneck_1d block's third convolution):
 * gw = d/dw of conv1d(relu(x * sc + sh)), gb = column sums of gy */
int lf_debug_conv1d_wgrad_pro(const float* x, const float* gy, const float* sc, const float* sh, float* gw, float* gb, int N, int H, int W,
                              int C, int axis, int dilation, float* scratch, void* stream);
/* lf_conv1d_bwd_data with the THREE-TENSOR epilogue of the network's last data gradient per block (ADD + MASK + BN-backward sums,
 * ERFNet.py:44-60 backward): gx = (conv1d^T(gy) + add_src) * [mask_src > 0]; stats receives the per-tile partial rows
 * [2][C][rows] (channel-major: element (kind, c, row)) = (sum gx, sum gx * aux) -- raw, as lf_bn_bwd_finalize consumes them.  Returns the number of rows, -1 on error. */
int lf_debug_conv1d_bwd_data_epi3(const float* gy, const float* w, const float* mask_src, const float* add_src, const float* aux,
                                  float* gx, float* stats, int N, int H, int W, int C, int axis, int dilation, float* scratch, void* stream);
/* one convolution launch with any epilogue flag set of csrc/lf_conv.h (1 ReLU, 2 mask by mask_src > 0, 4 + add_src, 8 BN forward sums,
 * 16 mask by aux * msc + msh > 0, 32 BN-backward sums over aux); transposed = 1: the data gradient's weights.  Returns the number of
 * statistics rows written (channel-major [2][C][rows]; BN forward sums: kind 1 = M2 about the row's own mean), 0 without a sums flag, -1 on error.  (tests/test_lean_gpu.py) */
int lf_debug_conv1d_epi(const float* src, const float* w, const float* bias, float* dst, int transposed, int epi, const float* mask_src,
                        const float* add_src, const float* aux, const float* msc, const float* msh, float* stats, int N, int H, int W, int C,
                        int axis, int dilation, float* scratch, void* stream);
int lf_debug_conv1d_wgrad_phases(const float* x, const float* gy, int N, int H, int W, int C, int axis, int dilation,
                                 float* scratch, unsigned long long* dbg, void* stream);
#ifdef __cplusplus
}
#endif
