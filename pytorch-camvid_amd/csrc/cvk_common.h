// Shared helpers for libcvk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cvk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void cvk_set_error(const char* fmt, ...);

#define CVK_CHECK_ARG(cond, ...)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            cvk_set_error(__VA_ARGS__);                  \
            return CVK_EINVAL;                           \
        }                                                \
    } while (0)

#define CVK_LAUNCH_RETURN(name)                                                  \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            cvk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                      \
        }                                                                        \
        return CVK_OK;                                                           \
    } while (0)

// Kernel-selection / timing switches read from the environment exist ONLY in the experiments build (`make experiments` ->
// lib/libcvk_exp.so, compiled with -DCVK_EXPERIMENTS; the tools/ timing scripts load it through CVK_LIB_PATH).  In the product
// library cvk_knob(name, default) is the constant `default`: libcvk.so never calls getenv and holds no CVK_* name, so a stray
// variable in a user's environment cannot change a kernel, let alone select one of the wrong-result ablation variants
// (tests/test_abi.py::test_product_library_reads_no_environment).
#ifdef CVK_EXPERIMENTS
#include <stdlib.h>
static inline int cvk_knob_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define cvk_knob(name, dflt) cvk_knob_env(name, dflt)
#else
#define cvk_knob(name, dflt) (dflt)
#endif

static inline bool cvk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int cvk_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so give every
// XCD one contiguous chunk of the logical tile order -> neighbouring tiles (shared halos / weight panels) hit the
// same 4 MiB L2.  Placement is a speed assumption only; any mapping is correct.
__device__ __forceinline__ int cvk_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- "amax blocks": the largest magnitude of a tensor, left in device memory by the pass that writes it (or by cvk_absmax_f32) ------------------
// One block = CVK_AMAX_SLOTS words, CVK_AMAX_STRIDE words (one 128-byte line) apart, each the atomicMax of fp32 bit patterns (non-negative floats
// order like their bits); the value is the maximum over the slots (cvk_amax_read).  A single word would be one hot L2 line: tens of thousands of
// short-lived waves reading and raising it serialise there (measured: the BN-apply pass 0.95 -> 2.2 ms with one word; 64 slots cost every
// consumer workgroup 64 scalar loads at its start instead: the two GEMMs 4.9 -> 6.1 ms; 8 slots + one publish per workgroup: both within noise).
// A wave (or workgroup: cvk_amax_publish_wg) publishes into the slot of its number, and only when it can raise it; the guard is a SNAPSHOT of
// the slot taken at the top of the kernel (cvk_amax_snapshot: its latency hides under the kernel's own loads; a stale snapshot is a lower bound,
// so skipping on it is safe).
// The caller zeroes the block (CVK_AMAX_WORDS words) first.  Every lane of a wave calls snapshot / publish.
#define CVK_AMAX_SLOTS 8
#define CVK_AMAX_STRIDE 32
#define CVK_AMAX_WORDS (CVK_AMAX_SLOTS * CVK_AMAX_STRIDE)
__device__ __forceinline__ unsigned* cvk_amax_slot(unsigned* block) {
    return block + (size_t)((blockIdx.x + blockIdx.y) % CVK_AMAX_SLOTS) * CVK_AMAX_STRIDE;
}
__device__ __forceinline__ unsigned cvk_amax_snapshot(unsigned* block) {
    return block != nullptr ? __atomic_load_n(cvk_amax_slot(block), __ATOMIC_RELAXED) : 0xFFFFFFFFu;
}
__device__ __forceinline__ void cvk_amax_publish_bits(unsigned m, unsigned* block, unsigned snapshot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m > snapshot) atomicMax(cvk_amax_slot(block), m);
}
__device__ __forceinline__ void cvk_amax_publish(float mx, unsigned* block, unsigned snapshot) {
    cvk_amax_publish_bits(__builtin_bit_cast(unsigned, mx) & 0x7FFFFFFFu, block, snapshot);
}
// one publish per 256-thread workgroup (kernels with thousands of short-lived workgroups); every thread calls it
__device__ __forceinline__ void cvk_amax_publish_wg(float mx, unsigned* block, unsigned snapshot) {
    __shared__ unsigned cvk_amax_wm[4];
    unsigned m = __builtin_bit_cast(unsigned, mx) & 0x7FFFFFFFu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0) cvk_amax_wm[(threadIdx.x >> 6) & 3] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a01 = cvk_amax_wm[0] > cvk_amax_wm[1] ? cvk_amax_wm[0] : cvk_amax_wm[1];
        const unsigned a23 = cvk_amax_wm[2] > cvk_amax_wm[3] ? cvk_amax_wm[2] : cvk_amax_wm[3];
        const unsigned bm = a01 > a23 ? a01 : a23;
        if (bm > snapshot) atomicMax(cvk_amax_slot(block), bm);
    }
}
// the block's value; `block` is wave-uniform: CVK_AMAX_SLOTS scalar loads
__device__ __forceinline__ unsigned cvk_amax_read(const unsigned* __restrict__ block) {
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < CVK_AMAX_SLOTS; ++i) {
        const unsigned v = block[i * CVK_AMAX_STRIDE];
        m = v > m ? v : m;
    }
    return m;
}
