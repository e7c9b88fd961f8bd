// LDS-DMA helpers shared by the kernels that count their own s_waitcnt vmcnt (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One 16-byte LDS-DMA per lane (global_load_lds_dwordx4): LDS destination = wave-uniform byte address in M0 + 16 * lane.
// Issued from inline asm so that hipcc's wait-count pass does not see it: hipcc orders every ds_read behind a tracked
// LDS-DMA with s_waitcnt vmcnt(0) when it cannot disambiguate the addresses, which serialises a multi-buffered loop.  The
// ordering is the kernel's: counted s_waitcnt vmcnt + s_barrier before the buffer is read.  For kernels that use no
// compiler-issued LDS-DMA (nothing of hipcc's lives in M0).
__device__ __forceinline__ void cvk_dma16(const void* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_byte_addr) : "memory");
}

__device__ __forceinline__ unsigned cvk_lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int N> __device__ __forceinline__ void cvk_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
