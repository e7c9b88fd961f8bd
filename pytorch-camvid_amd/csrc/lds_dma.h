// LDS-DMA helpers shared by the kernels that count their own s_waitcnt vmcnt (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One 16-byte LDS-DMA per lane (global_load_lds_dwordx4): LDS destination = wave-uniform byte address in M0 + 16 * lane.
// Issued from inline asm so that hipcc's wait-count pass does not see it: hipcc orders every ds_read behind a tracked
// LDS-DMA with s_waitcnt vmcnt(0) when it cannot disambiguate the addresses, which serialises a multi-buffered loop.  The
// ordering is the kernel's: counted s_waitcnt vmcnt + s_barrier before the buffer is read.  For kernels that use no
// compiler-issued LDS-DMA and no dynamically indexed register arrays (nothing of hipcc's lives in M0): M0 cannot be named in
// the clobber list — hipcc rejects it as a reserved register ("may not be preserved across the asm statement", tried in
// round 3) — so the kernels that include this header keep every register array fully unrolled / constant-indexed, which
// tools/isa_order.py shows as the absence of v_movrel / s_movrel in their ISA.
__device__ __forceinline__ void cvk_dma16(const void* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_byte_addr) : "memory");
}

__device__ __forceinline__ unsigned cvk_lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int N> __device__ __forceinline__ void cvk_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Barrier that retires an LDS buffer which the next instructions refill: the wave's own LDS reads are drained first.  hipcc
// sinks the MFMAs consuming the last fragment reads below a raw s_barrier, so those reads would still be queued when another
// wave's refill is issued (csrc/conv_bf16s.hip lds_retire_barrier has the failure this caused).
__device__ __forceinline__ void cvk_lds_retire_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
