// conv_tile.h — constants and helpers shared by the implicit-GEMM conv kernels (conv3x3.hip, wino.hip).
#pragma once
#include "cvk_common.h"

namespace {

constexpr int BK = 32;        // K slice per LDS stage (floats)
constexpr int LDT = BK + 4;   // padded LDS row: 144 B -> the 16 rows of a ds_read_b128 lane group hit 16 distinct slots
constexpr unsigned OOB = 0x80000000u;  // buffer offset beyond num_records (< 2 GiB by contract): the load returns 0

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// Exact unsigned division by a runtime constant d >= 1 without branches: q = floor(t / d) for t*d < 2^32.
// magic = floor(2^32/d) + 1 needs 33 bits when d == 1, so it is kept as (hi ? 2^32 : 0) + lo with hi_mask = all-ones iff d == 1.
struct FastDiv {
    unsigned lo, hi_mask;
    __host__ __device__ explicit FastDiv(unsigned d) {
        const unsigned long long m = 0x100000000ULL / d + 1ULL;
        lo = (unsigned)m;
        hi_mask = (m >> 32) ? 0xFFFFFFFFu : 0u;
    }
    __device__ __forceinline__ unsigned div(unsigned t) const { return __umulhi(t, lo) + (t & hi_mask); }
};

// byte offset, forced out of range (-> the buffer load returns 0) when !ok; pure ALU so the staging code stays branch-free
__device__ __forceinline__ unsigned oob_unless(bool ok, unsigned off) { return off | ((unsigned)(!ok) << 31); }

}  // namespace
