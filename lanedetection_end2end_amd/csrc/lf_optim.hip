// Fused multi-tensor Adam step (the step right after the hot path; the reference builds
// torch.optim.Adam(lr=1e-4, weight_decay=wd) in BEV/Networks/utils.py:411-420 and calls optimizer.step() at
// BEV/main.py:266).  ONE launch updates all 226 parameter tensors: a device table maps each workgroup to a
// (tensor, 4096-element chunk); arithmetic follows torch.optim.Adam (L2 weight decay folded into the gradient,
// bias-corrected moments, eps added after the square root).
#include "lf_common.h"

struct LfAdamTensor { float* p; const float* g; float* m; float* v; long numel; };

namespace {
constexpr int ADAM_CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_kernel(const LfAdamTensor* __restrict__ tensors, const int2* __restrict__ work,
                                                  float lr, float beta1, float beta2, float eps, float weight_decay,
                                                  float bc1, float bc2_sqrt, float grad_scale) {
    const int2 wk = work[blockIdx.x];
    const LfAdamTensor t = tensors[wk.x];
    const long base = (long)wk.y * ADAM_CHUNK;
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
}  // namespace

extern "C" {

int lf_adam_chunk(void) { return ADAM_CHUNK; }

// tensors_dev: n LfAdamTensor records {p, g, m, v, numel} (5 x 8 bytes each); work_dev: nblocks int2 (tensor, chunk).
// step >= 1 is the 1-based step count AFTER this update; grad_scale multiplies every gradient first (1/world for
// summed data-parallel gradients, 1 otherwise).
int lf_adam_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_scale, void* stream) {
    LF_REQUIRE(tensors_dev && work_dev && nblocks > 0 && step >= 1, "lf_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    LF_CHECK_LAUNCH("adam_step");
    return 0;
}

}  // extern "C"
