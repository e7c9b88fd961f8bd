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

}  // namespace
