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
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { OPT_ADAM = 0, OPT_SGD = 1, OPT_RMSPROP = 2 };
struct OptScalars { float a, b, c, d, e, f; };      // per-optimizer constants (see update())

// One element.  OPT_ADAM: a = step_size (lr / bias correction 1), b = beta1, c = beta2, d = eps, e = weight decay, f = sqrt(bias
// correction 2); torch.optim.Adam's arithmetic (lerp for the first moment, eps added after the square root).  OPT_SGD: a = lr,
// b = momentum, e = weight decay (dampening 0, no Nesterov; a zero buffer reproduces torch's first step).  OPT_RMSPROP: a = lr,
// b = momentum, c = alpha, d = eps, e = weight decay (not centered).  Record fields: m = exp_avg / momentum buffer, v =
// exp_avg_sq / square average.
template <int OP>
__device__ __forceinline__ void update(float& p, float g, float& m, float& v, const OptScalars& k) {
    if (k.e != 0.f) g = fmaf(k.e, p, g);
    if constexpr (OP == OPT_ADAM) {
        m = fmaf(1.f - k.b, g - m, m);
        v = fmaf(k.c, v, (1.f - k.c) * g * g);
        p = p - k.a * (m / (sqrtf(v) / k.f + k.d));
    } else if constexpr (OP == OPT_SGD) {
        m = fmaf(k.b, m, g);
        p = p - k.a * m;
    } else {
        v = fmaf(k.c, v, (1.f - k.c) * g * g);
        m = fmaf(k.b, m, g / (sqrtf(v) + k.d));
        p = p - k.a * m;
    }
}

// One workgroup = one (tensor, 4096-element chunk).  A full chunk is 16 elements per thread as four float4 groups with ALL
// their loads issued before the first update (round 4: the first version walked the chunk in 16 dependent scalar rounds --
// 188 us per step where the 58 MB it moves take 10); the gradient is a view into the flat gradient buffer at an arbitrary
// element offset, so its vector form is taken only when the pointer happens to be 16-byte aligned.
template <int OP>
__global__ __launch_bounds__(256) void opt_kernel(const LfAdamTensor* __restrict__ tensors, const int2* __restrict__ work,
                                                 OptScalars k, float lr, int parity, float grad_scale) {
    const int2 wk = work[blockIdx.x];
    const LfAdamTensor t = tensors[wk.x];
    const long base = (long)wk.y * ADAM_CHUNK;
    if constexpr (OP == OPT_ADAM) {
        const long step = t.step[parity] + 1;               // 1-based count of THIS tensor's updates, including this one
        if (wk.y == 0 && threadIdx.x == 0) const_cast<LfAdamTensor*>(tensors)[wk.x].step[parity ^ 1] = step;
        k.a = lr / (float)(1.0 - pow((double)k.b, (double)step));
        k.f = (float)sqrt(1.0 - pow((double)k.c, (double)step));
    }
    const bool gvec = ((reinterpret_cast<unsigned long long>(t.g) & 15ull) == 0ull);
    if (base + ADAM_CHUNK <= t.numel) {
        float P[4][4], G[4][4], M[4][4], V[4][4];
        auto ld = [](float (&d)[4], const float* s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(s);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        };
        auto st = [](float* d, const float (&s)[4]) { *reinterpret_cast<f32x4*>(d) = f32x4{s[0], s[1], s[2], s[3]}; };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = base + j * 1024 + threadIdx.x * 4;
            ld(P[j], t.p + e);
            ld(M[j], t.m + e);
            if constexpr (OP != OPT_SGD) ld(V[j], t.v + e);
            else { V[j][0] = V[j][1] = V[j][2] = V[j][3] = 0.f; }
            if (gvec) ld(G[j], t.g + e);
            else { G[j][0] = t.g[e]; G[j][1] = t.g[e + 1]; G[j][2] = t.g[e + 2]; G[j][3] = t.g[e + 3]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = base + j * 1024 + threadIdx.x * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) update<OP>(P[j][q], G[j][q] * grad_scale, M[j][q], V[j][q], k);
            st(t.p + e, P[j]);
            st(t.m + e, M[j]);
            if constexpr (OP != OPT_SGD) st(t.v + e, V[j]);
        }
        return;
    }
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {     // a tensor's last (or only) chunk
        const long e = base + i;
        if (e >= t.numel) break;
        float p = t.p[e], m = t.m[e], v = OP != OPT_SGD ? t.v[e] : 0.f;
        update<OP>(p, t.g[e] * grad_scale, m, v, k);
        t.p[e] = p;
        t.m[e] = m;
        if constexpr (OP != OPT_SGD) t.v[e] = v;
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
    const OptScalars k = {lr, momentum, 0.f, 0.f, weight_decay, 0.f};
    hipLaunchKernelGGL(opt_kernel<OPT_SGD>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, k, lr, 0, grad_scale);
    LF_CHECK_LAUNCH("sgd_step");
    return 0;
}
int lf_rmsprop_step(const void* tensors_dev, const void* work_dev, int nblocks, float lr, float alpha, float eps, float momentum,
                    float weight_decay, float grad_scale, void* stream) {
    LF_REQUIRE(tensors_dev && work_dev && nblocks > 0, "lf_rmsprop_step: bad arguments");
    const OptScalars k = {lr, momentum, alpha, eps, weight_decay, 0.f};
    hipLaunchKernelGGL(opt_kernel<OPT_RMSPROP>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, k, lr, 0, grad_scale);
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
    const OptScalars k = {0.f, beta1, beta2, eps, weight_decay, 1.f};      // a (step size) and f (bias correction 2) per tensor, in the kernel
    hipLaunchKernelGGL(opt_kernel<OPT_ADAM>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const LfAdamTensor*)tensors_dev,
                       (const int2*)work_dev, k, lr, parity, grad_scale);
    LF_CHECK_LAUNCH("adam_step");
    return 0;
}

}  // extern "C"
