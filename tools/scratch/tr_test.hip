#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    short h[1024], o[256]; for (int i = 0; i < 1024; ++i) h[i] = (short)i;
    short *d, *e; hipMalloc(&d, 2048); hipMalloc(&e, 512); hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int want = (l & 15) + j * 16 + (l >> 4) * 64; if (o[l * 4 + j] != want) ++bad; }
    printf("mismatches vs lds[(l&15) + j*16 + (l>>4)*64]: %d\n", bad);
    for (int l = 0; l < 64; l += 17) printf("lane %2d: %d %d %d %d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}
