// conv_tile.h — constants and helpers shared by the implicit-GEMM conv kernels (conv3x3.hip, wino.hip).
#pragma once
#include "cvk_common.h"

namespace {

constexpr int BK = 32;        // K slice per LDS stage (floats)
constexpr int LDT = BK + 4;   // padded LDS row: 144 B -> the 16 rows of a ds_read_b128 lane group hit 16 distinct slots
constexpr unsigned OOB = 0x80000000u;  // buffer offset beyond num_records (< 2 GiB by construction): the load returns 0

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// Buffer resource over the tail [first, total) (in floats) of a tensor.  Every workgroup addresses its operand through a
// window that starts just before the first pixel it can touch, so 32-bit byte offsets only have to span one workgroup's
// working set and tensors may be arbitrarily larger than the 2 GiB a single resource can address.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t window_rsrc(const float* base, size_t first, size_t total) {
    const size_t bytes = first < total ? (total - first) * 4 : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + first), 0, (int)(bytes < 0x7FFFFFFFu ? bytes : 0x7FFFFFFFu), 0x00020000);
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

// dw[co][tap][ci] = sum_s slab[s][co][tap*Cin_pad + ci], fixed order (bitwise reproducible)
__global__ void k_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dw, int splits, int Cout, int Cin,
                               int Cin_pad, size_t slab_stride) {
    const size_t total = (size_t)Cout * 9 * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t ct = i / Cin;  // co*9 + tap
        const float* p = slab + ct * Cin_pad + ci;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            s0 += p[(size_t)(s + 0) * slab_stride];
            s1 += p[(size_t)(s + 1) * slab_stride];
            s2 += p[(size_t)(s + 2) * slab_stride];
            s3 += p[(size_t)(s + 3) * slab_stride];
        }
        for (; s < splits; ++s) s0 += p[(size_t)s * slab_stride];
        dw[i] = (s0 + s1) + (s2 + s3);
    }
}


// ---- launch-shape heuristics --------------------------------------------------------------------------------------
// Split the pixel dimension of the weight-grad so that the grid is close to a whole number of rounds over the
// 256 CUs (equal blocks => efficiency = blocks / (256 * ceil(blocks/256))), keeping >= 16 K-slices per block.
inline int choose_splits(int tiles, int M) {
    const int max_splits = M / (BK * 16) > 0 ? M / (BK * 16) : 1;
    int best = 1;
    double best_eff = -1.0;
    for (int s = 1; s <= max_splits && s <= 2048; ++s) {
        const long blocks = (long)tiles * s;
        if (blocks > 4096 && s > 1) break;
        if (blocks < 512 && s < max_splits) continue;  // want >= 2 co-resident blocks per CU to cover barrier bubbles
        const long rounds = (blocks + 255) / 256;
        const double eff = (double)blocks / (256.0 * rounds);
        if (eff > best_eff + 0.02) {
            best_eff = eff;
            best = s;
        }
    }
    return best;
}

struct WgradPlan {
    int bm, bn, tilesM, tilesN, splits, chunk;
};

inline WgradPlan plan_wgrad(int M, int Cin_pad, int Cout) {
    WgradPlan p;
    const int Ktot = 9 * Cin_pad;
    if (Cout > 64) { p.bm = 128; p.bn = 128; }
    else if (Cout > 32) { p.bm = 64; p.bn = Ktot <= 64 ? 64 : 128; }   // 64x64: the 3-channel stem (36 columns)
    else { p.bm = 32; p.bn = 256; }
    p.tilesM = cvk_cdiv(Cout, p.bm);
    p.tilesN = cvk_cdiv(Ktot, p.bn);
    p.splits = choose_splits(p.tilesM * p.tilesN, M);
    p.chunk = cvk_cdiv(cvk_cdiv(M, p.splits), BK) * BK;
    p.splits = cvk_cdiv(M, p.chunk);
    return p;
}

}  // namespace
