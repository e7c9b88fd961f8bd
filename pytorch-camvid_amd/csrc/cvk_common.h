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
