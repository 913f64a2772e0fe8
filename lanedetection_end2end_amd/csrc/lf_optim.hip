// Fused multi-tensor Adam step (the step right after the hot path; the reference builds
// torch.optim.Adam(lr=1e-4, weight_decay=wd) in BEV/Networks/utils.py:411-420 and calls optimizer.step() at
// BEV/main.py:266).  ONE launch updates all 226 parameter tensors: a device table maps each workgroup to a
// (tensor, 4096-element chunk); arithmetic follows torch.optim.Adam (L2 weight decay folded into the gradient,
// bias-corrected moments, eps added after the square root).  The step count -- hence the bias correction -- is PER TENSOR, as in
// torch.optim.Adam: the reference's pretrained schedule switches heads under one optimizer (BEV/main.py get_flags), so
// decoder.output_conv gets its first gradient when the other tensors are at step k.  The counts live in the device table,
// double-buffered: a launch reads step[parity] of its tensor and the tensor's first workgroup writes step[parity ^ 1].
#include "lf_common.h"

struct LfAdamTensor { float* p; const float* g; float* m; float* v; long numel; long step[2]; };

namespace {
constexpr int ADAM_CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_kernel(const LfAdamTensor* __restrict__ tensors, const int2* __restrict__ work,
                                                  float lr, float beta1, float beta2, float eps, float weight_decay,
                                                  int parity, float grad_scale) {
    const int2 wk = work[blockIdx.x];
    const LfAdamTensor t = tensors[wk.x];
    const long base = (long)wk.y * ADAM_CHUNK;
    const long step = t.step[parity] + 1;                   // 1-based count of THIS tensor's updates, including this one
    if (wk.y == 0 && threadIdx.x == 0) const_cast<LfAdamTensor*>(tensors)[wk.x].step[parity ^ 1] = step;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const float step_size = lr / bc1;
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {
        const long k = base + i;
        if (k >= t.numel) break;
        float g = t.g[k] * grad_scale;
        const float p = t.p[k];
        if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
        const float m = fmaf(1.f - beta1, g - t.m[k], t.m[k]);          // lerp, as torch does
        const float v = fmaf(beta2, t.v[k], (1.f - beta2) * g * g);
        t.m[k] = m;
        t.v[k] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        t.p[k] = p - step_size * (m / denom);
    }
}
// torch.optim.SGD(momentum, dampening 0, no Nesterov): g += wd * p; buf = momentum * buf + g; p -= lr * buf.  A zero-initialised
// buffer reproduces torch's first step (buf = g).  Record field m = momentum buffer.
__global__ __launch_bounds__(256) void sgd_kernel(const LfAdamTensor* __restrict__ tensors, const int2* __restrict__ work,
                                                 float lr, float momentum, float weight_decay, float grad_scale) {
    const int2 wk = work[blockIdx.x];
    const LfAdamTensor t = tensors[wk.x];
    const long base = (long)wk.y * ADAM_CHUNK;
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {
        const long k = base + i;
        if (k >= t.numel) break;
        float g = t.g[k] * grad_scale;
        const float p = t.p[k];
        if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
        const float b = fmaf(momentum, t.m[k], g);
        t.m[k] = b;
        t.p[k] = p - lr * b;
    }
}

// torch.optim.RMSprop(alpha, eps, momentum, not centered): g += wd * p; sq = alpha * sq + (1 - alpha) * g^2;
// buf = momentum * buf + g / (sqrt(sq) + eps); p -= lr * buf.  Record fields v = square average, m = momentum buffer.
__global__ __launch_bounds__(256) void rmsprop_kernel(const LfAdamTensor* __restrict__ tensors, const int2* __restrict__ work,
                                                     float lr, float alpha, float eps, float momentum, float weight_decay,
                                                     float grad_scale) {
    const int2 wk = work[blockIdx.x];
    const LfAdamTensor t = tensors[wk.x];
    const long base = (long)wk.y * ADAM_CHUNK;
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {
        const long k = base + i;
        if (k >= t.numel) break;
        float g = t.g[k] * grad_scale;
        const float p = t.p[k];
        if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
        const float sq = fmaf(alpha, t.v[k], (1.f - alpha) * g * g);
        t.v[k] = sq;
        const float b = fmaf(momentum, t.m[k], g / (sqrtf(sq) + eps));
        t.m[k] = b;
        t.p[k] = p - lr * b;
    }
}
}  // namespace

extern "C" {

int lf_adam_chunk(void) { return ADAM_CHUNK; }

// The other two optimizers the reference's define_optim builds (BEV/Networks/utils.py:411-420): SGD(momentum 0.9) and
// RMSprop(momentum 0.9), one launch each over the same record / work tables as lf_adam_step (the step slots are unused).
int lf_sgd_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float momentum, float weight_decay,
                float grad_scale, void* stream) {
    LF_REQUIRE(tensors_dev && work_dev && nblocks > 0, "lf_sgd_step: bad arguments");
    hipLaunchKernelGGL(sgd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, lr, momentum, weight_decay, grad_scale);
    LF_CHECK_LAUNCH("sgd_step");
    return 0;
}
int lf_rmsprop_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float alpha, float eps, float momentum,
                    float weight_decay, float grad_scale, void* stream) {
    LF_REQUIRE(tensors_dev && work_dev && nblocks > 0, "lf_rmsprop_step: bad arguments");
    hipLaunchKernelGGL(rmsprop_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, lr, alpha, eps, momentum, weight_decay, grad_scale);
    LF_CHECK_LAUNCH("rmsprop_step");
    return 0;
}

// tensors_dev: n LfAdamTensor records {p, g, m, v, numel, step[2]} (7 x 8 bytes each); work_dev: nblocks int2 (tensor, chunk).
// parity (0 / 1): which of a record's two step slots holds the number of updates the tensor has received so far; the launch
// leaves that number + 1 in the other slot, so the caller alternates parity and never uploads a count.  grad_scale multiplies
// every gradient first (1/world for summed data-parallel gradients, 1 otherwise).
int lf_adam_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int parity, float grad_scale, void* stream) {
    LF_REQUIRE(tensors_dev && work_dev && nblocks > 0 && (parity == 0 || parity == 1), "lf_adam_step: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, lr, beta1, beta2, eps, weight_decay, parity, grad_scale);
    LF_CHECK_LAUNCH("adam_step");
    return 0;
}

}  // extern "C"
