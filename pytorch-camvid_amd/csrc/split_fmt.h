// Split-plane operand formats of the opt-in split-operand GEMMs (csrc/split3.hip consumes them, csrc/wino2d.hip's transforms write them).
//
//   fmt 3: three bf16 terms  x = x1 + x2 + x3          (8 + 8 + 8 mantissa bits, six cross-products per fp32 product)
//   fmt 2: two fp16 terms    s * x = h1 + h2           (11 + 11 mantissa bits, three cross-products per fp32 product) with a power-of-two
//          scale s per transform index that places the plane inside fp16's range: s = 2^e, e = 15 - (floor(log2 amax) + 1) - c_i - c_j,
//          where amax is the largest magnitude of the tensor the transform READS (an exact device-side maximum in an "amax block" of
//          device memory, cvk_common.h: cvk_absmax_f32, or the pass that wrote the tensor) and
//          2^c_i >= the absolute row sum of row i of the 1-D transform matrix — so |s * value| < 2^15 for every element by construction,
//          whatever the data.  Elements more than ~2^17 below the plane's bound lose relative (never absolute) precision as h2 goes
//          subnormal: their absolute error stays below 2^-40 of the bound.  The GEMM epilogues multiply by 2^-(e_a + e_b): exact.
//
// Layout (both): 16-bit [xi][C/32][term][Rpad][32], the 16-byte chunk (c % 32) / 8 of a 64-byte row at position chunk ^ (2 * ((row >> 2) & 1)).
#pragma once
#include <hip/hip_runtime.h>
#include "cvk_common.h"

// kinds of 1-D transform matrix a plane was produced with
enum { CVK_SPLIT_KIND_B = 0, CVK_SPLIT_KIND_G = 1, CVK_SPLIT_KIND_A = 2 };

struct CvkSplitTab {
    unsigned long long c8;  // byte i = 64 + c[i], c[i] = ceil(log2(sum_k |M[i][k]|)) of the 1-D transform matrix M (rows i < nt); packed so that a
                            // kernel extracts c[i] of a run-time i with a shift (no indexed register array: see lds_dma.h on M0)
    int nt;                 // points per dimension (6 | 8): xi = i * nt + j
};
__host__ __device__ inline int cvk_split_tab_c(const CvkSplitTab& t, int i) { return (int)((t.c8 >> (8 * i)) & 0xFFull) - 64; }

// table of (tile, kind); defined in wino2d.hip from the transform routines themselves
CvkSplitTab cvk_split_tab(int tile, int kind);

// exponent of the scale of transform index (i, j) for a source tensor whose largest magnitude has the fp32 bit pattern amax_bits
__host__ __device__ inline int cvk_split_exp(unsigned amax_bits, int ci, int cj) {
    if ((amax_bits & 0x7FFFFFFFu) == 0u) return 0;                      // an all-zero tensor: any scale serves
    const int ea = (int)((amax_bits >> 23) & 0xFFu) - 127;              // amax < 2^(ea + 1)
    const int e = 15 - (ea + 1) - ci - cj;                              // >= -126 for any finite amax (ci + cj <= 12)
    return e < -126 ? -126 : (e > 126 ? 126 : e);                       // the upper clamp only meets tensors below 2^-100: they lose precision, not range
}
__host__ __device__ inline float cvk_pow2f(int e) {                    // 2^e for -126 <= e <= 127
    union { unsigned u; float f; } v;
    v.u = (unsigned)(127 + e) << 23;
    return v.f;
}
__device__ __forceinline__ int cvk_split_exp_xi(const unsigned* __restrict__ amax, const CvkSplitTab& tab, int xi) {
    const int i = xi / tab.nt, j = xi - i * tab.nt;
    return cvk_split_exp(cvk_amax_read(amax), cvk_split_tab_c(tab, i), cvk_split_tab_c(tab, j));
}
// the two factors 2^h, 2^(es - h) of 2^es, -252 <= es <= 252, applied one after the other: every intermediate lies between the operand and the
// result, so nothing overflows or underflows that the exact product would not
struct CvkUnscale { float a, b; };
__device__ __forceinline__ CvkUnscale cvk_unscale(int es) {
    const int h = es / 2;
    return CvkUnscale{cvk_pow2f(h), cvk_pow2f(es - h)};
}
