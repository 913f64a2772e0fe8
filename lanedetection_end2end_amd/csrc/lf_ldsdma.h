// LDS-DMA helpers shared by the bf16 kernels that stage operands through hand-ordered LDS rings (lf_conv.hip: the ring /
// whole-line tap-GEMMs and the 16-channel weight gradient; lf_wgrad_ro.hip: the read-once weight gradient).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr unsigned LF_OOB = 0xffff0000u;      // byte offset beyond every tensor the launchers admit (< 4 GiB - 64 KiB)

// One LDS-DMA instruction: 64 lanes x 16 bytes, lane-linear at LDS byte address `lds_addr` (wave-uniform, through M0).
// Inline asm on purpose: with the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc tracks the asynchronous LDS writes and
// puts s_waitcnt vmcnt(0) in front of the first LDS read that may alias them -- every K-step, which drains the ring and leaves
// ONE step in flight; __syncthreads() does the same through its release fence.  The rings are ordered by an explicit
// s_waitcnt vmcnt(N) + bare s_barrier instead.  (Compiler-issued waits stay correct: they can only over-wait.)
typedef int i32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma16(i32x4s rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    // (s_nop 0: the wait state between the SALU write of M0 and the LDS-DMA instruction that reads it -- nothing pads the inside of
    // an asm string)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4s make_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long ad = (unsigned long long)base;
    i32x4s r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ad);
    r.y = __builtin_amdgcn_readfirstlane((int)((ad >> 32) & 0xffffu));       // stride 0, no swizzle
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
// my DMA of the oldest stage has landed (N instructions of younger stages may be in flight) and my LDS reads have retired
template <int N> __device__ __forceinline__ void wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

}  // namespace
