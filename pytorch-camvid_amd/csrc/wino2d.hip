// libcvk — 2-D Winograd F(4x4, 3x3) for the deep fp32 layers (>= 256 channels on both sides at <= 90 x 120 pixels).
//
// Replaces: nn.Conv2d(k=3, pad=1) forward and data-grad of BasicConv2d (reference models/unet.py:5-17, bwd of train.py:131)
// on the layers where 36 transform-domain GEMMs  M_xi[tiles][Cout] = V_xi[tiles][Cin] * U_xi[Cout][Cin]^T  beat the 1-D
// F(4,3) kernels of wino4.hip:  2.25 multiplies per output and input channel instead of 4.5 (direct: 9).  The price is
// three HBM passes (input transform 1 read + 2.25 writes, product planes 2.25 writes + 2.25 reads, output 1 write), which
// is why only channel-heavy layers take this path (cost model: engine.py wino2d_pays).
//   V = B^T d B   (6 x 6 input tile, zero padded)      U = G g G^T      y = A^T [ sum_ci U (.) V ] A   (4 x 4 outputs)
// with the points 0, +-1, +-2, inf of wino4.hip.  Rounding: 2-3e-6 relative L2 per layer in fp32 (1-D F(4,3): 6-8e-7, direct
// 2.5e-7; tests/test_gpu_blocks.py derives its tolerance from a numpy restatement of exactly these transforms).
//
// The GEMM has NO address arithmetic in its K loop: both operands are K-contiguous rows, staged by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction) into a ring of three stages, so the vector ALU — which shares issue
// slots with the fp32 MFMA (tools/micro/mfma_peak.hip) — only runs ds_reads.
#include "cvk_common.h"
#include "lds_dma.h"
#include "split_fmt.h"

namespace {

constexpr int W2_BN = 128;          // output columns (channels) per workgroup
// tiles per statistics partial / output-transform block: 16, or 4 for layers with few tiles (16 would leave the 45x60 and
// 22x30 levels with fewer blocks than CUs)
__host__ __device__ inline int w2_tb(int T) { return T >= 8192 ? 16 : 4; }

// ---- 1-D transforms (applied along rows, then along columns) ---------------------------------------------------------------
// B^T d
template <typename T>
__host__ __device__ __forceinline__ void w2_bt(const T* d, T* v) {
    const T a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    v[1] = a + b;
    v[2] = a - b;
    v[3] = c + e;
    v[4] = c - e;
    v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A^T m
template <typename T>
__device__ __forceinline__ void w2_at(const T* m, T* y) {
    const T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}
// G g
__host__ __device__ __forceinline__ void w2_g(const float* g, float* u) {
    const float t = g[0] + g[2];
    u[0] = 0.25f * g[0];
    u[1] = (-1.f / 6.f) * (t + g[1]);
    u[2] = (-1.f / 6.f) * (t - g[1]);
    const float q = (1.f / 24.f) * g[0] + (1.f / 6.f) * g[2];
    u[3] = q + (1.f / 12.f) * g[1];
    u[4] = q - (1.f / 12.f) * g[1];
    u[5] = g[2];
}

// A (6 x 4) = transpose of A^T above: e0 = d0, e1 = d0+d1+d2+d3, e2 = d0-d1+d2-d3, e3 = d0+2d1+4d2+8d3, e4 = d0-2d1+4d2-8d3, e5 = d3
template <typename T>
__host__ __device__ __forceinline__ void w2_a(const T* d, T* e) {
    const T s02 = d[0] + d[2], s13 = d[1] + d[3], t02 = d[0] + 4.f * d[2], t13 = 2.f * d[1] + 8.f * d[3];
    e[0] = d[0];
    e[1] = s02 + s13;
    e[2] = s02 - s13;
    e[3] = t02 + t13;
    e[4] = t02 - t13;
    e[5] = d[3];
}

// ---- F(6x6, 3x3): points 0, +-1, +-2, +-1/2, inf (Lavin & Gray) — 64 products per 36 outputs = 1.78 multiplies per output and
// input channel (F(4x4): 2.25), and 1.78x instead of 2.25x the activation in transform-domain planes.  Rounding in fp32: ~2x
// that of F(4x4,3x3) (2.7e-6 / 5.1e-6 relative L2 at 64 / 256 input channels against 1.4e-6 / 2.7e-6, numpy restatement).
// B^T d (8 -> 8)
template <typename T>
__host__ __device__ __forceinline__ void w6_bt(const T* d, T* v) {
    v[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
    v[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
    const T a1 = d[2] + d[6] - 4.25f * d[4], b1 = d[1] + d[5] - 4.25f * d[3];
    v[1] = a1 + b1;
    v[2] = a1 - b1;
    const T a2 = d[6] + 0.25f * d[2] - 1.25f * d[4], b2 = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    v[3] = a2 + b2;
    v[4] = a2 - b2;
    const T a3 = d[6] + 4.f * d[2] - 5.f * d[4], b3 = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    v[5] = a3 + b3;
    v[6] = a3 - b3;
}
// A^T m (8 -> 6)
template <typename T>
__device__ __forceinline__ void w6_at(const T* m, T* y) {
    const T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
    y[0] = m[0] + s12 + s34 + s56;
    y[1] = d12 + 2.f * d34 + 0.5f * d56;
    y[2] = s12 + 4.f * s34 + 0.25f * s56;
    y[3] = d12 + 8.f * d34 + 0.125f * d56;
    y[4] = s12 + 16.f * s34 + 0.0625f * s56;
    y[5] = d12 + 32.f * d34 + 0.03125f * d56 + m[7];
}
// G g (3 -> 8)
__host__ __device__ __forceinline__ void w6_g(const float* g, float* u) {
    const float t = g[0] + g[2];
    u[0] = g[0];
    u[1] = (-2.f / 9.f) * (t + g[1]);
    u[2] = (-2.f / 9.f) * (t - g[1]);
    const float q = (1.f / 90.f) * g[0] + (2.f / 45.f) * g[2];
    u[3] = q + (1.f / 45.f) * g[1];
    u[4] = q - (1.f / 45.f) * g[1];
    const float r = (32.f / 45.f) * g[0] + (8.f / 45.f) * g[2];
    u[5] = r + (16.f / 45.f) * g[1];
    u[6] = r - (16.f / 45.f) * g[1];
    u[7] = g[2];
}
// A d (6 -> 8): the transpose of the output transform (weight-grad: dy tiles)
template <typename T>
__host__ __device__ __forceinline__ void w6_a(const T* d, T* e) {
    const T ev = d[0] + d[2] + d[4], od = d[1] + d[3] + d[5];
    e[0] = d[0];
    e[1] = ev + od;
    e[2] = ev - od;
    const T ev2 = d[0] + 4.f * d[2] + 16.f * d[4], od2 = 2.f * d[1] + 8.f * d[3] + 32.f * d[5];
    e[3] = ev2 + od2;
    e[4] = ev2 - od2;
    const T ev3 = d[0] + 0.25f * d[2] + 0.0625f * d[4], od3 = 0.5f * d[1] + 0.125f * d[3] + 0.03125f * d[5];
    e[5] = ev3 + od3;
    e[6] = ev3 - od3;
    e[7] = d[5];
}
// G^T p (8 -> 3)
__device__ __forceinline__ void w6_gt(const float* p, float* w) {
    const float s12 = p[1] + p[2], d12 = p[1] - p[2], s34 = p[3] + p[4], d34 = p[3] - p[4], s56 = p[5] + p[6], d56 = p[5] - p[6];
    w[0] = p[0] - (2.f / 9.f) * s12 + (1.f / 90.f) * s34 + (32.f / 45.f) * s56;
    w[1] = -(2.f / 9.f) * d12 + (1.f / 45.f) * d34 + (16.f / 45.f) * d56;
    w[2] = -(2.f / 9.f) * s12 + (2.f / 45.f) * s34 + (8.f / 45.f) * s56 + p[7];
}
// G^T p (6 -> 3) of F(4,3)
__device__ __forceinline__ void w2_gt(const float* m, float* w) {
    const float s12 = m[1] + m[2], d12 = m[2] - m[1], s34 = m[3] + m[4], d34 = m[3] - m[4];
    w[0] = 0.25f * m[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    w[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
    w[2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + m[5];
}

typedef float f32x2v __attribute__((ext_vector_type(2)));

// Tile traits: MT x MT outputs per tile, NT = MT + 2 points per dimension, NX = NT^2 transform indices (= batched GEMMs); VT = the
// channel vector one thread of a transform kernel owns (a whole tile lives in its registers: 36 x 4 or 64 x 2 floats).
template <int MT> struct W2T;
template <> struct W2T<4> {
    static constexpr int NT = 6, NX = 36, VW = 4;
    typedef f32x4 VT;
    template <typename T> static __host__ __device__ __forceinline__ void bt(const T* d, T* v) { w2_bt(d, v); }
    template <typename T> static __host__ __device__ __forceinline__ void a(const T* d, T* e) { w2_a(d, e); }
    static __host__ __device__ __forceinline__ void g(const float* x, float* u) { w2_g(x, u); }
    static __device__ __forceinline__ void gt(const float* x, float* w) { w2_gt(x, w); }
    // A^T[row][col]
    static __device__ __forceinline__ constexpr float at(int r, int c) {
        return c == 0 ? (r == 0 ? 1.f : 0.f) : c == 5 ? (r == 3 ? 1.f : 0.f)
             : c == 1 ? 1.f : c == 2 ? ((r & 1) ? -1.f : 1.f)
             : c == 3 ? (float)(1 << r) : ((r & 1) ? -(float)(1 << r) : (float)(1 << r));
    }
};
template <> struct W2T<6> {
    static constexpr int NT = 8, NX = 64, VW = 2;
    typedef f32x2v VT;
    template <typename T> static __host__ __device__ __forceinline__ void bt(const T* d, T* v) { w6_bt(d, v); }
    template <typename T> static __host__ __device__ __forceinline__ void a(const T* d, T* e) { w6_a(d, e); }
    static __host__ __device__ __forceinline__ void g(const float* x, float* u) { w6_g(x, u); }
    static __device__ __forceinline__ void gt(const float* x, float* w) { w6_gt(x, w); }
    static __device__ __forceinline__ constexpr float at(int r, int c) {
        return c == 0 ? (r == 0 ? 1.f : 0.f) : c == 7 ? (r == 5 ? 1.f : 0.f)
             : c == 1 ? 1.f : c == 2 ? ((r & 1) ? -1.f : 1.f)
             : c == 3 ? (float)(1 << r) : c == 4 ? ((r & 1) ? -(float)(1 << r) : (float)(1 << r))
             : c == 5 ? 1.f / (float)(1 << r) : ((r & 1) ? -1.f / (float)(1 << r) : 1.f / (float)(1 << r));
    }
};

// ---- input transform: x [N,H,W,C] -> V [NX][T][C], T = N * ceil(H/MT) * ceil(W/MT); one thread = one tile x VW channels ------
// Rows T <= t < Tpad (weight-grad: the tile index is the GEMM depth, padded to whole K slices) are written as zeros.
// ---- split planes (csrc/split_fmt.h; consumed by csrc/split3.hip) written straight from a transform's store loop ----------------------------
// 16-bit [xi][C/32][term][Rpad][32]: FMT 3 = channel c of row t of transform index xi as x1 + x2 + x3 (three bf16 roundings of the remainder),
// FMT 2 = 2^e(xi) * value as h1 + h2 (two fp16 roundings; e from the source tensor's absolute maximum, split_fmt.h); the 16-byte chunk
// (c % 32) / 8 of a 64-byte row at position chunk ^ (2 * ((t >> 2) & 1)).  One call stores VW consecutive channels (c % VW == 0).
struct SplitDst {
    unsigned short* base;       // element (xi = 0, this thread's channel slice, term 0, row t, swizzled chunk, c % 8)
    size_t term, xstride;       // elements between two terms / two transform indices
    unsigned amax;              // FMT 2: bit pattern of the source tensor's largest magnitude
    CvkSplitTab tab;
};
template <int VW, int FMT>
__device__ __forceinline__ SplitDst split_dst(void* S, int C, int Rpad, int t, int c, const unsigned* amax, const CvkSplitTab& tab) {
    const int ncs = C >> 5, cs = c >> 5, cl = c & 31;
    const int pos = (cl >> 3) ^ (((t >> 2) & 1) << 1);
    SplitDst d;
    d.term = (size_t)Rpad * 32;
    d.xstride = (size_t)ncs * FMT * d.term;
    d.base = reinterpret_cast<unsigned short*>(S) + ((size_t)cs * FMT * Rpad + t) * 32 + pos * 8 + (cl & 7);
    d.amax = FMT == 2 ? cvk_amax_read(amax) : 0u;
    d.tab = tab;
    return d;
}
__device__ __forceinline__ unsigned short w2_bf16_rne(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// one 16-bit term of r (and r -= its value): bf16 for FMT 3, fp16 for FMT 2 (both round to nearest even; the remainder is exact in fp32)
template <int FMT>
__device__ __forceinline__ unsigned short split_term(float& r) {
    if (FMT == 2) {
        const _Float16 h = (_Float16)r;
        r -= (float)h;
        return __builtin_bit_cast(unsigned short, h);
    }
    const unsigned short b = w2_bf16_rne(r);
    r -= __builtin_bit_cast(float, (unsigned)b << 16);
    return b;
}
// i, j: the transform index' row / column (compile-time constants in the unrolled store loops)
template <int VW, int FMT, typename VT>
__device__ __forceinline__ void split_store(const SplitDst& d, int xi, int i, int j, VT v) {
    float r[VW];
    const float sc = FMT == 2 ? cvk_pow2f(cvk_split_exp(d.amax, cvk_split_tab_c(d.tab, i), cvk_split_tab_c(d.tab, j))) : 1.f;
#pragma unroll
    for (int q = 0; q < VW; ++q) r[q] = FMT == 2 ? v[q] * sc : v[q];
    unsigned short* p = d.base + (size_t)xi * d.xstride;
#pragma unroll
    for (int k = 0; k < FMT; ++k) {
        unsigned short b[VW];
#pragma unroll
        for (int q = 0; q < VW; ++q) b[q] = split_term<FMT>(r[q]);
        if (VW == 2) *reinterpret_cast<unsigned*>(p) = (unsigned)b[0] | ((unsigned)b[1] << 16);
        else {
            typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2w*>(p) = u32x2w{(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[VW > 2 ? 2 : 0] | ((unsigned)b[VW > 2 ? 3 : 0] << 16)};
        }
        p += d.term;
    }
}

// one value: row r, channel c of transform index (i, j) of split planes with Rpad rows and C channels
template <int FMT>
__device__ __forceinline__ void split_store1(void* S, int C, int Rpad, int i, int j, int r, int c, float v, unsigned amax, const CvkSplitTab& tab) {
    const int ncs = C >> 5, cs = c >> 5, cl = c & 31;
    const int pos = (cl >> 3) ^ (((r >> 2) & 1) << 1);
    unsigned short* p = reinterpret_cast<unsigned short*>(S) + ((((size_t)(i * tab.nt + j) * ncs + cs) * FMT) * Rpad + r) * 32 + pos * 8 + (cl & 7);
    const size_t term = (size_t)Rpad * 32;
    if (FMT == 2) v *= cvk_pow2f(cvk_split_exp(amax, cvk_split_tab_c(tab, i), cvk_split_tab_c(tab, j)));
#pragma unroll
    for (int k = 0; k < FMT; ++k) p[k * term] = split_term<FMT>(v);
}

template <int MT, int SPL = 0>       // SPL: 0 = fp32 planes, 3 / 2 = split planes of that format (split_fmt.h)
__global__ __launch_bounds__(256) void k_w2d_input(const float* __restrict__ X, float* __restrict__ V, int H, int W, int C,
                                                  int th, int tw, int T, int Tpad, const unsigned* __restrict__ amax, CvkSplitTab tab) {
    typedef W2T<MT> TR;
    typedef typename TR::VT VT;
    constexpr int NT = TR::NT, NX = TR::NX, VW = TR::VW;
    const int cvn = C / VW;
    // workgroups are dealt to the XCDs in contiguous chunks of tiles: vertically neighbouring tiles share two input rows, and a
    // tile row is tw tiles — many workgroups — away
    const long idx = (long)cvk_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    int t, c;
    if (SPL) {
        // split planes are [slice of 32 channels][row t][32]: a thread group of 32 / VW lanes covers one row of ONE slice and consecutive
        // groups take consecutive rows, so that a wave's stores of one (xi, term) are one contiguous run (4 or 8 rows x 64 B) — with the
        // fp32 mapping (all channels of a tile side by side) they were 64-byte pieces Tpad rows apart and the pass ran at half its rate
        constexpr int LPR = 32 / VW;                              // lanes per row of a slice
        const long rowid = idx / LPR;
        const int cs = (int)(rowid / Tpad);
        t = (int)(rowid % Tpad);
        c = cs * 32 + (int)(idx % LPR) * VW;
        if (cs >= C / 32) return;
    } else {
        t = (int)(idx / cvn); c = (int)(idx % cvn) * VW;
    }
    if (t >= Tpad) return;
    const VT zero = {};
    SplitDst sd = {};
    if (SPL) sd = split_dst<VW, SPL ? SPL : 3>(V, C, Tpad, t, c, amax, tab);
    if (t >= T) {
        for (int xi = 0; xi < NX; ++xi) {
            if (SPL) split_store<VW, SPL ? SPL : 3>(sd, xi, 0, 0, zero);
            else *reinterpret_cast<VT*>(V + ((size_t)xi * Tpad + t) * C + c) = zero;
        }
        return;
    }
    const int n = t / (th * tw), r = t - n * th * tw, ty = r / tw, tx = r - ty * tw;
    const int y0 = MT * ty - 1, x0 = MT * tx - 1;
    const float* const xb = X + (size_t)n * H * W * C + c;
    VT w[NT][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        VT d[NT], v[NT];
        const int xx = x0 + j;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int yy = y0 + i;
            const bool ok = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
            d[i] = ok ? *reinterpret_cast<const VT*>(xb + ((size_t)yy * W + xx) * C) : zero;
        }
        TR::bt(d, v);
#pragma unroll
        for (int i = 0; i < NT; ++i) w[i][j] = v[i];
    }
    const size_t plane = (size_t)Tpad * C;
    float* const vb = V + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        VT v[NT];
        TR::bt(w[i], v);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (SPL) split_store<VW, SPL ? SPL : 3>(sd, i * NT + j, i, j, v[j]);
            else *reinterpret_cast<VT*>(vb + (size_t)(i * NT + j) * plane) = v[j];
        }
    }
}

// ---- weight transform: w [Co][3][3][Ci] -> U [NX][Co][Ci]; one thread = one (co, ci) --------------------------------------------
template <int MT, int SPL = 0>      // SPL 3 / 2: U as split planes [xi][Ci/32][term][Co padded to 128][32] (rows beyond Co are zeroed by the caller)
__device__ __forceinline__ void w2d_weight_body(const float* __restrict__ Wt, float* __restrict__ U, int Co, int Ci, unsigned vblock, unsigned nblocks,
                                                unsigned amax = 0u, const CvkSplitTab& tab = CvkSplitTab{}) {
    typedef W2T<MT> TR;
    constexpr int NT = TR::NT;
    const size_t total = (size_t)Co * Ci;
    for (size_t i = (size_t)vblock * 256 + threadIdx.x; i < total; i += (size_t)nblocks * 256) {
        const int ci = (int)(i % Ci);
        const size_t co = i / Ci;
        float g[3][3], a[NT][3];
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = Wt[(co * 9 + k) * Ci + ci];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {                 // along ky
            const float col[3] = {g[0][kx], g[1][kx], g[2][kx]};
            float u[NT];
            TR::g(col, u);
#pragma unroll
            for (int p = 0; p < NT; ++p) a[p][kx] = u[p];
        }
#pragma unroll
        for (int p = 0; p < NT; ++p) {                   // along kx
            float u[NT];
            TR::g(a[p], u);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                if (SPL) split_store1<SPL ? SPL : 3>(U, Ci, (Co + 127) / 128 * 128, p, q, (int)co, ci, u[q], amax, tab);
                else U[(size_t)(p * NT + q) * total + i] = u[q];
            }
        }
    }
}
template <int MT, int SPL = 0>
__global__ __launch_bounds__(256) void k_w2d_weight(const float* __restrict__ Wt, float* __restrict__ U, int Co, int Ci, const unsigned* __restrict__ amax,
                                                   CvkSplitTab tab) {
    w2d_weight_body<MT, SPL>(Wt, U, Co, Ci, blockIdx.x, gridDim.x, SPL == 2 ? cvk_amax_read(amax) : 0u, tab);
}

// ---- data-grad filter straight from the forward weights: U [NX][Ci][Co] = G w'[ci][.][.][co] G^T with w'[ci][r][s][co] =
// w[co][2-r][2-s][ci] (rotated by 180 degrees, channels exchanged) — no packed copy in between.  A workgroup transposes a
// 32 (co) x 32 (ci) tile of all nine taps through LDS: reads run along ci, writes along co.
template <int MT, int SPL = 0>      // SPL 3 / 2: split planes [xi][Co/32][term][Ci padded to 128][32]
__device__ __forceinline__ void w2d_weight_dgrad_body(const float* __restrict__ Wt, float* __restrict__ U, int Co, int Ci, int bx, int by, float (*t)[32][33],
                                                      unsigned amax = 0u, const CvkSplitTab& tab = CvkSplitTab{}) {
    typedef W2T<MT> TR;
    constexpr int NT = TR::NT;
    const int ci0 = bx * 32, co0 = by * 32;
    float ld[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) {                       // all 36 loads of a thread in flight at once
        const int e = threadIdx.x + 256 * j;
        const int ci = e & 31, co = (e >> 5) & 31, tap = e >> 10;
        ld[j] = (co0 + co < Co && ci0 + ci < Ci) ? Wt[((size_t)(co0 + co) * 9 + tap) * Ci + ci0 + ci] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 36; ++j) {
        const int e = threadIdx.x + 256 * j;
        t[e >> 10][(e >> 5) & 31][e & 31] = ld[j];
    }
    __syncthreads();
    const size_t total = (size_t)Co * Ci;
    for (int p = threadIdx.x; p < 1024; p += 256) {
        const int co = p & 31, ci = p >> 5;
        if (co0 + co >= Co || ci0 + ci >= Ci) continue;
        float a[NT][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {                 // along ky (rotated: kernel row r reads tap row 2 - r)
            const float col[3] = {t[8 - kx][co][ci], t[5 - kx][co][ci], t[2 - kx][co][ci]};
            float u[NT];
            TR::g(col, u);
#pragma unroll
            for (int q = 0; q < NT; ++q) a[q][kx] = u[q];
        }
        const size_t o = (size_t)(ci0 + ci) * Co + co0 + co;
#pragma unroll
        for (int q = 0; q < NT; ++q) {                   // along kx
            float u[NT];
            TR::g(a[q], u);
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                if (SPL) split_store1<SPL ? SPL : 3>(U, Co, (Ci + 127) / 128 * 128, q, r, ci0 + ci, co0 + co, u[r], amax, tab);
                else U[(size_t)(q * NT + r) * total + o] = u[r];
            }
        }
    }
}
template <int MT, int SPL = 0>
__global__ __launch_bounds__(256) void k_w2d_weight_dgrad(const float* __restrict__ Wt, float* __restrict__ U, int Co, int Ci, const unsigned* __restrict__ amax,
                                                         CvkSplitTab tab) {
    __shared__ float t[9][32][33];
    w2d_weight_dgrad_body<MT, SPL>(Wt, U, Co, Ci, blockIdx.x, blockIdx.y, t, SPL == 2 ? cvk_amax_read(amax) : 0u, tab);
}

// All 2-D Winograd filter transforms of a step (forward and data-grad filters, both tile sizes) in ONE launch: job j owns the blocks
// [first[j], first[j + 1]); the jobs travel by value in the kernel arguments.
struct W2WJobsDev { const float* w[CVK_WT_BATCH_MAX]; float* out[CVK_WT_BATCH_MAX]; int Co[CVK_WT_BATCH_MAX], Ci[CVK_WT_BATCH_MAX], kind[CVK_WT_BATCH_MAX];
                    unsigned first[CVK_WT_BATCH_MAX + 1]; int n; };      // kind = tile (4 | 6) + 16 * dgrad
__global__ __launch_bounds__(256) void k_w2d_weight_batch(const W2WJobsDev jobs) {
    __shared__ float t[9][32][33];
    int j = 0;
    while (j + 1 < jobs.n && blockIdx.x >= jobs.first[j + 1]) ++j;
    const unsigned vb = blockIdx.x - jobs.first[j], nb = jobs.first[j + 1] - jobs.first[j];
    const int Co = jobs.Co[j], Ci = jobs.Ci[j], kind = jobs.kind[j];
    if (kind & 16) {
        const int gx = (Ci + 31) / 32;
        if ((kind & 15) == 4) w2d_weight_dgrad_body<4>(jobs.w[j], jobs.out[j], Co, Ci, (int)(vb % gx), (int)(vb / gx), t);
        else w2d_weight_dgrad_body<6>(jobs.w[j], jobs.out[j], Co, Ci, (int)(vb % gx), (int)(vb / gx), t);
    } else {
        if ((kind & 15) == 4) w2d_weight_body<4>(jobs.w[j], jobs.out[j], Co, Ci, vb, nb);
        else w2d_weight_body<6>(jobs.w[j], jobs.out[j], Co, Ci, vb, nb);
    }
}

// ---- batched GEMM  D_xi[T][ldd] = A_xi[T][K] * B_xi[Nn][K]^T,  xi = 0..35 ------------------------------------------------------
// Workgroup: BM x 128 tile, BM/32 waves in a (BM/64) x 2 grid of 64 x 64 wave tiles (2 x 2 MFMA 32x32x2 blocks, 64
// accumulator registers).  LDS: NSTG stages of (BM + 128) rows x 128 bytes; 16-byte chunk c of row r sits at position
// c ^ ((r >> 1) & 7) (applied through the DMA source address), so the sixteen rows of a ds_read_b128 phase hit sixteen
// different 16-byte bank groups.  A lane's b128 holds k = 8s + 4h + {0..3} of its row: four MFMAs per read, the same K
// permutation on both operands.  Per K slice: counted wait for the slice's own DMA pieces, one barrier, DMA of slice
// + NSTG - 1.  Used as <128, 32, 2, 2>: 4 waves, 64 KiB, two workgroups per CU — co-residency (one workgroup's prologue and
// store epilogue under the other's MFMAs) is worth more than the larger tile: <256, 32, 3> (8 waves, 144 KiB, one per CU) was
// 3-9 % slower on every layer, <128, 16, 3> (three per CU) and <256, 16, 2> (two per CU) within +-3 % (tools/bench_conv.py w2d).
// Tail: the tiles of the last, partial round of workgroups (ids >= split_start) are cut into f K-ranges, one workgroup
// each; range p writes plane p of D (plane stride part_stride) and the output transform adds the planes in a fixed order.
template <int BM, int BK, int NSTG, int WPS, int ABL = 0>      // tile rows, K slice (floats), LDS stages, waves per SIMD the kernel is built for;
// ABL (experiments build only, WRONG results): 1 no barrier, 2 no DMA in the loop, 3 no stores, 4 no LDS reads, 5 MFMA only, 6 every DMA re-reads the tile's first slice (cache-hot);
// 8, 10, 11, 12 (correct results): other placements of the DMA issue inside the slice
__global__ __launch_bounds__(BM * 2, WPS) void k_w2d_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ D, int T, int Tpad, int Nn, int K, int ldd,
                                                         int tilesM, int tilesN, int split_start, int f, size_t part_stride, int stagger) {
    constexpr int NW = BM / 32, ROWS = BM + W2_BN, ROWB = BK * 4, RPP = 1024 / ROWB, NCH = ROWB / 16;   // rows per 1 KiB DMA piece, chunks per row
    constexpr int PIECES = ROWS / RPP, PPW = PIECES / NW, STAGE = ROWS * ROWB, NS = BK / 8;
    constexpr int SWS = NCH == 8 ? 1 : 2;               // swizzle: chunk c of row r at c ^ ((r >> SWS) & (NCH-1))
    static_assert(PIECES % NW == 0 && (NSTG == 2 || NSTG == 3) && (BK == 16 || BK == 32), "stage geometry");
    static_assert(NSTG * STAGE * (WPS * 4 / NW) <= 160 * 1024, "LDS budget of the co-resident workgroups");
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // full-round workgroups and tail workgroups are dealt to the XCDs separately (both in contiguous logical chunks)
    const int nK = K / BK;
    int id, kb = 0, ke = nK;
    if ((int)blockIdx.x < split_start) {
        id = cvk_xcd_remap(blockIdx.x, split_start);
    } else {
        const int q = cvk_xcd_remap(blockIdx.x - split_start, gridDim.x - split_start);
        const int part = q % f;
        id = split_start + q / f;
        kb = part * nK / f;
        ke = (part + 1) * nK / f;
        D += (size_t)part * part_stride;
    }
    const int tn = id % tilesN, tm = (id / tilesN) % tilesM, xi = id / (tilesN * tilesM);
    A += (size_t)xi * Tpad * K;                           // V planes carry Tpad rows (zero beyond T), the product planes T
    B += (size_t)xi * Nn * K;
    D += (size_t)xi * T * ldd;
    const int row0 = tm * BM, col0 = tn * W2_BN;

    // DMA mapping: piece wave*PPW + q = LDS rows RPP*piece + lane/NCH; rows < BM are A rows, the rest B rows (clamped: rows
    // beyond the matrix re-read its last row, their products are never stored)
    const float* src[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int row = (wave * PPW + q) * RPP + lane / NCH;
        const int chunk = (lane % NCH) ^ ((row >> SWS) & (NCH - 1));
        src[q] = (row < BM ? A + (size_t)min(row0 + row, T - 1) * K : B + (size_t)min(col0 + row - BM, Nn - 1) * K) + chunk * 4;
    }
    auto issue = [&](int ks, int buf) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) cvk_dma16(src[q] + ks * BK, smem_addr + buf * STAGE + (wave * PPW + q) * 1024);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses: row block base + lane row r, chunk (2s + h) ^ swizzle(r); s enters as an XOR of bits 5..
    const int lane_off = r * ROWB + ((h ^ ((r >> SWS) & (NCH - 1))) << 4);
    const int a_off = wm * 64 * ROWB + lane_off, b_off = (BM + wn * 64) * ROWB + lane_off;

    // Co-resident workgroups start together and run equal tiles, so their prologues (DMA round trip before the first MFMA) and
    // store epilogues would coincide for the whole launch; the second wave of workgroups (blockIdx 256..511: the second one on
    // each CU) starts half a tile late, and every later workgroup inherits the phase shift (it starts when a predecessor
    // finishes): 1-4 % per layer.  (Tried without effect on this kernel: pinned fragment prefetch one k-group ahead — two
    // waves per SIMD already hide the LDS latency — and 4- or 5-deep rings of 16-float slices.)
    // Only where the assumption holds: `stagger` = number of CUs when the grid has at least two full-tile workgroups per CU
    // (else 0, set by the host: CVK_W2D_NO_STAGGER=1 turns it off for A/B timing, e.g. beside RCCL kernels under data-parallel
    // runs); workgroups [stagger, 2 * stagger) are the second ones dispatched to each CU.  Full-round tiles only.
    if (stagger > 0 && (int)blockIdx.x >= stagger && (int)blockIdx.x < 2 * stagger && (int)blockIdx.x < split_start)
        for (int i = 0; i < (ke - kb) * (BK / 8) / 8; ++i) __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int d = 0; d < NSTG - 1; ++d) issue(min(kb + d, ke - 1), d);
    int buf = 0;
    for (int ks = kb; ks < ke; ++ks) {
        if (ABL != 2 && ABL != 5) cvk_wait_vm<(NSTG - 2) * PPW>();                  // this wave's pieces of slice ks (later slices may be in flight)
        if (ABL != 1 && ABL != 5) cvk_lds_retire_barrier();                         // slice ks complete; stage of slice ks-1 free (its reads drained)
        // (buf + NSTG - 1) % NSTG.  Two stages: every wait above is vmcnt(0), so the last slice simply issues nothing (round 6; before, it
        // re-read itself to keep the count uniform: one slice of redundant DMA per tile — 1/8 of the tile's traffic at K = 256 — and the
        // stores below waited for it to land)
        const bool more = NSTG > 2 || ks + 1 < ke;
        const int ksn = ABL == 6 ? kb : min(ks + NSTG - 1, ke - 1), bufn = buf == 0 ? NSTG - 1 : buf - 1;
        // Round 6: the slice's DMA pieces are NOT issued here in one burst (ABL 12 keeps that form): a wave issues in order, and eight DMA instructions
        // ahead of the first fragment reads kept its MFMAs waiting for their issue plus the LDS latency.  Now the reads of group 0 go first, half
        // of the pieces follow (their issue covers the reads' latency), the other half after the reads of group 1: -1.2 ... -2.4 % on the 26
        // launches of the headline step (tools/bench_w2d_gemm.py, two boxes; two pieces per group and all eight after group 0's reads were tried too).
        constexpr bool SPLIT_ISSUE = PPW % 2 == 0 && NS >= 2 && !(ABL == 2 || ABL == 5 || ABL == 8 || ABL == 10 || ABL == 11 || ABL == 12);
        if ((ABL == 12 || (!SPLIT_ISSUE && ABL != 2 && ABL != 5 && ABL < 8)) && more) issue(ksn, bufn);
        // round 6: the wave is in its MFMA phase until the end of the slice — raise its priority over the co-resident workgroup's waves that are
        // issuing DMA / waiting at their barrier (interleaved A/B on one box: 6.80 -> 6.75 ms for the 26 launches, 0.7186 -> 0.7235 executed)
        __builtin_amdgcn_s_setprio(1);
        const char* const st = smem + buf * STAGE;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (ABL == 4 || ABL == 5) { a[i] = f32x4{acc[i][0][0], acc[i][0][1], 1.f, 2.f}; asm volatile("" : "+v"(a[i])); }
                else a[i] = *reinterpret_cast<const f32x4*>(st + ((a_off + i * 32 * ROWB) ^ (s << 5)));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ABL == 4 || ABL == 5) { b[j] = f32x4{acc[0][j][2], acc[0][j][3], 1.f, 2.f}; asm volatile("" : "+v"(b[j])); }
                else b[j] = *reinterpret_cast<const f32x4*>(st + ((b_off + j * 32 * ROWB) ^ (s << 5)));
            }
            // ABL 8 / 10 / 11 / 12 (correct results): other placements of the slice's DMA pieces (two per group / all after group 0's reads / 2+3+3 / burst at the top)
            if (ABL == 8 && more) {
#pragma unroll
                for (int q = 2 * s; q < 2 * s + 2; ++q) cvk_dma16(src[q] + ksn * BK, smem_addr + bufn * STAGE + (wave * PPW + q) * 1024);
            }
            if (ABL == 10 && more && s == 0) {
#pragma unroll
                for (int q = 0; q < PPW; ++q) cvk_dma16(src[q] + ksn * BK, smem_addr + bufn * STAGE + (wave * PPW + q) * 1024);
            }
            if (ABL == 11 && more && s < 3) {
#pragma unroll
                for (int q = (s == 0 ? 0 : 3 * s - 1); q < 3 * s + 2; ++q) cvk_dma16(src[q] + ksn * BK, smem_addr + bufn * STAGE + (wave * PPW + q) * 1024);
            }
            if (SPLIT_ISSUE && more && s < 2) {
#pragma unroll
                for (int q = s * (PPW / 2); q < (s + 1) * (PPW / 2); ++q) cvk_dma16(src[q] + ksn * BK, smem_addr + bufn * STAGE + (wave * PPW + q) * 1024);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        buf = buf == NSTG - 1 ? 0 : buf + 1;
    }
    cvk_wait_vm<0>();                                     // the redundant tail DMAs land before the workgroup's LDS is released
    // Stores.  Full-width tiles (all of this launch's when Nn is a multiple of 128) go out as raw buffer stores: the row bound is the
    // descriptor's range check (row >= T -> byte offset >= T * ldd * 4 -> dropped by the hardware) and the register's row offset a
    // scalar, so a store costs no vector instruction besides itself (the checked form below: a compare, a mask and a 64-bit add each).
    if (ABL != 3 && col0 + W2_BN <= Nn && ((size_t)T + BM) * (size_t)ldd * 4 < ((size_t)1 << 32)) {
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)D, 0, (int)((size_t)T * ldd * 4), 0x00020000);
        const unsigned voff = ((unsigned)(row0 + wm * 64 + 4 * h) * (unsigned)ldd + (unsigned)(col0 + wn * 64 + r)) * 4u;     // < 2^32 by the test above
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned soff = (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * (unsigned)ldd * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][e]), dr, voff + j * 128, soff, 0);
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (ABL == 3 ? (row < T && col < Nn && acc[i][j][e] == 123.456f) : (row < T && col < Nn)) D[(size_t)row * ldd + col] = acc[i][j][e];
            }
        }
}

// ================================================================================================ weight-grad
// dW = G^T [ sum_tiles (A dy A^T) (.) (B^T x B) ] G:  per transform index one GEMM  P_xi[Cout][Cin] = E_xi^T V_xi  whose depth
// is the tile index (reference: the weight gradient of nn.Conv2d, bwd of train.py:131).
// dy [N,H,W,ld] -> E [NX][Tpad][C]; one thread = one MT x MT tile x VW channels (pixels beyond the frame count as zero)
template <int MT>
__global__ __launch_bounds__(256) void k_w2d_dy(const float* __restrict__ DY, int ld, float* __restrict__ E, int H, int W, int C,
                                               int th, int tw, int T, int Tpad) {
    typedef W2T<MT> TR;
    typedef typename TR::VT VT;
    constexpr int NT = TR::NT, NX = TR::NX, VW = TR::VW;
    const int cvn = C / VW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int t = (int)(idx / cvn), c = (int)(idx % cvn) * VW;
    if (t >= Tpad) return;
    const VT zero = {};
    const size_t plane = (size_t)Tpad * C;
    float* const eb = E + (size_t)t * C + c;
    if (t >= T) {
        for (int xi = 0; xi < NX; ++xi) *reinterpret_cast<VT*>(eb + (size_t)xi * plane) = zero;
        return;
    }
    const int n = t / (th * tw), r = t - n * th * tw, ty = r / tw, tx = r - ty * tw;
    const float* const db = DY + (size_t)n * H * W * ld + c;
    VT w[NT][MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        VT d[MT], e[NT];
        const int xx = MT * tx + j;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int yy = MT * ty + i;
            d[i] = (yy < H && xx < W) ? *reinterpret_cast<const VT*>(db + ((size_t)yy * W + xx) * ld) : zero;
        }
        TR::a(d, e);
#pragma unroll
        for (int i = 0; i < NT; ++i) w[i][j] = e[i];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        VT e[NT];
        TR::a(w[i], e);
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<VT*>(eb + (size_t)(i * NT + j) * plane) = e[j];
    }
}

// dy -> BOTH transforms of the backward pass in one launch: V' = B^T dy B (the data-grad's GEMM operand: the tile with its one-pixel
// halo) and E = A dy A^T (the weight-grad's: the tile's own MT x MT pixels).  The thread that has just transformed a tile for V'
// re-reads its inner pixels (L1 / L2 hits) for E: dy crosses the fabric once instead of twice, and one launch replaces two.
template <int MT, int SPL = 0, int FMT = 3>      // SPL bit 0: V' as split planes, bit 1: E as split planes (both with Tpad rows), of format FMT
__global__ __launch_bounds__(256) void k_w2d_dy_both(const float* __restrict__ DY, int ld, float* __restrict__ Vp, float* __restrict__ E, int H,
                                                    int W, int C, int th, int tw, int T, int Tpad, int TpadE, const unsigned* __restrict__ amax,
                                                    CvkSplitTab tabB, CvkSplitTab tabA) {
    // Tpad: rows of the V' planes (and of the launch), TpadE <= Tpad: rows of the E planes (fp32 E for the fp32 weight-grad GEMM keeps its
    // 32-row padding beside 256-row split V' planes)
    typedef W2T<MT> TR;
    typedef typename TR::VT VT;
    constexpr int NT = TR::NT, NX = TR::NX, VW = TR::VW;
    const int cvn = C / VW;
    const long idx = (long)cvk_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    int t, c;
    if (SPL) {          // slice-major thread order (see k_w2d_input)
        constexpr int LPR = 32 / VW;
        const long rowid = idx / LPR;
        const int cs = (int)(rowid / Tpad);
        t = (int)(rowid % Tpad);
        c = cs * 32 + (int)(idx % LPR) * VW;
        if (cs >= C / 32) return;
    } else {
        t = (int)(idx / cvn); c = (int)(idx % cvn) * VW;
    }
    if (t >= Tpad) return;
    const VT zero = {};
    const size_t plane = (size_t)Tpad * C, planeE = (size_t)TpadE * C;
    float* const vb = Vp + (size_t)t * C + c;
    float* const eb = E + (size_t)t * C + c;
    SplitDst sv = {}, se = {};
    if (SPL & 1) sv = split_dst<VW, FMT>(Vp, C, Tpad, t, c, amax, tabB);
    if (SPL & 2) se = split_dst<VW, FMT>(E, C, TpadE, t, c, amax, tabA);
    if (t >= T) {
        for (int xi = 0; xi < NX; ++xi) {
            if (SPL & 1) split_store<VW, FMT>(sv, xi, 0, 0, zero); else *reinterpret_cast<VT*>(vb + (size_t)xi * plane) = zero;
            if (t < TpadE) { if (SPL & 2) split_store<VW, FMT>(se, xi, 0, 0, zero); else *reinterpret_cast<VT*>(eb + (size_t)xi * planeE) = zero; }
        }
        return;
    }
    const int n = t / (th * tw), r = t - n * th * tw, ty = r / tw, tx = r - ty * tw;
    const float* const db = DY + (size_t)n * H * W * ld + c;
    {   // V' (as k_w2d_input)
        const int y0 = MT * ty - 1, x0 = MT * tx - 1;
        VT w[NT][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            VT d[NT], v[NT];
            const int xx = x0 + j;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int yy = y0 + i;
                const bool ok = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
                d[i] = ok ? *reinterpret_cast<const VT*>(db + ((size_t)yy * W + xx) * ld) : zero;
            }
            TR::bt(d, v);
#pragma unroll
            for (int i = 0; i < NT; ++i) w[i][j] = v[i];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            VT v[NT];
            TR::bt(w[i], v);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (SPL & 1) split_store<VW, FMT>(sv, i * NT + j, i, j, v[j]);
                else *reinterpret_cast<VT*>(vb + (size_t)(i * NT + j) * plane) = v[j];
            }
        }
    }
    {   // E (as k_w2d_dy)
        VT w[NT][MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            VT d[MT], e[NT];
            const int xx = MT * tx + j;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int yy = MT * ty + i;
                d[i] = (yy < H && xx < W) ? *reinterpret_cast<const VT*>(db + ((size_t)yy * W + xx) * ld) : zero;
            }
            TR::a(d, e);
#pragma unroll
            for (int i = 0; i < NT; ++i) w[i][j] = e[i];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            VT e[NT];
            TR::a(w[i], e);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (SPL & 2) split_store<VW, FMT>(se, i * NT + j, i, j, e[j]);
                else *reinterpret_cast<VT*>(eb + (size_t)(i * NT + j) * planeE) = e[j];
            }
        }
    }
}

// batched GEMM over the tile index:  D_xi[Mm][Nn] (plane `part`) = sum_{k in part's range} A_xi[k][Mm]^T * B_xi[k][Nn]
// Both operands are depth-major (a depth row = lda / ldb contiguous floats).  Workgroup: 128 x 128 tile, four waves of
// 64 x 64; LDS: two stages of 32 depth rows x (128 + 128) floats, copied by LDS-DMA as they lie (1 KiB = two depth rows of one
// operand).  A wave's two MFMA row blocks take interleaved rows (2*rho + i) and its column blocks interleaved columns, so one
// ds_read_b64 at depth row 2q + h delivers a lane's A values of both row blocks, another its B values: 32 reads per 64
// MFMAs, and the results of a lane are column pairs (8-byte stores).  The depth is cut into f ranges (grid.y) when the
// 36 * tiles workgroups alone would not fill the chip; the planes are added by k_w2d_wgrad_out in a fixed order.
#ifndef CVK_TN_SPLIT_ISSUE
#define CVK_TN_SPLIT_ISSUE 1
#endif
template <int ABL = 0>       // ABL (experiments build, WRONG results): 1 no barrier, 2 no DMA in the loop, 3 no stores, 4 no LDS reads
__global__ __launch_bounds__(256, 2) void k_w2d_gemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                       float* __restrict__ D, int Kp, int Mm, int Nn, int tilesM, int tilesN,
                                                       int f, int NX) {
    constexpr int STAGE = 32 * 256 * 4, PPW = 8;        // 32 KiB per stage: depth rows [k][A 128 | B 128]... kept as two 16 KiB halves
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int id = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int part = id % f, tile = id / f;
    const int tn = tile % tilesN, tm = (tile / tilesN) % tilesM, xi = tile / (tilesN * tilesM);
    const int nK = Kp / 32, kb = part * nK / f, ke = (part + 1) * nK / f;
    A += (size_t)xi * Kp * lda + tm * 128;
    B += (size_t)xi * Kp * ldb + tn * 128;
    D += ((size_t)part * NX + xi) * Mm * Nn;

    // DMA: piece p = wave*8 + q; pieces 0..15 = A depth rows 2p, 2p+1 (LDS bytes [0, 16 KiB)), 16..31 = B (LDS [16, 32 KiB));
    // lane = depth row lane/32 of the pair, floats 4*(lane%32) .. +3 of the 128.  Columns beyond the matrix read the
    // neighbouring bytes of the plane (in bounds: the planes carry a slack row); their products are never stored.
    const float* src[PPW];
    int sstep[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int p = wave * PPW + q, pa = p & 15;
        const int krow = pa * 2 + (lane >> 5), col = (lane & 31) * 4;
        src[q] = p < 16 ? A + (size_t)krow * lda + col : B + (size_t)krow * ldb + col;
        sstep[q] = p < 16 ? 32 * lda : 32 * ldb;
    }
    auto issue = [&](int ks, int buf) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) cvk_dma16(src[q] + (size_t)ks * sstep[q], smem_addr + buf * STAGE + (wave * PPW + q) * 1024);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int a_off = h * 512 + (wm * 64 + 2 * r) * 4, b_off = 16384 + h * 512 + (wn * 64 + 2 * r) * 4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    if (kb < ke) issue(kb, 0);
    int buf = 0;
    for (int ks = kb; ks < ke; ++ks) {
        if (ABL != 2) cvk_wait_vm<0>();
        if (ABL != 1) cvk_lds_retire_barrier();
        const int ksn = ABL == 5 ? kb : min(ks + 1, ke - 1);          // ABL 5: every DMA re-reads the first slice (cache-hot)
        if (!CVK_TN_SPLIT_ISSUE && ABL != 2) issue(ksn, buf ^ 1);
        __builtin_amdgcn_s_setprio(1);                   // MFMA phase of the slice (as k_w2d_gemm, round 6: 3.21 -> 3.14 ms for the 13 launches, 0.761 -> 0.778)
        const char* const st = smem + buf * STAGE;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x2 a, b;
            if (ABL == 4) { a = f32x2{acc[0][0][0], 1.f}; b = f32x2{acc[0][0][1], 2.f}; asm volatile("" : "+v"(a), "+v"(b)); }
            else {
                a = *reinterpret_cast<const f32x2*>(st + a_off + q * 1024);
                b = *reinterpret_cast<const f32x2*>(st + b_off + q * 1024);
            }
            if (CVK_TN_SPLIT_ISSUE && ABL != 2 && (q == 0 || q == 4)) {          // the next slice's DMA pieces behind the first fragment reads, in two halves (as k_w2d_gemm)
#pragma unroll
                for (int p = (q ? PPW / 2 : 0); p < (q ? PPW : PPW / 2); ++p)
                    cvk_dma16(src[p] + (size_t)ksn * sstep[p], smem_addr + (buf ^ 1) * STAGE + (wave * PPW + p) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        buf ^= 1;
    }
    cvk_wait_vm<0>();
    const int col = tn * 128 + wn * 64 + 2 * r;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = tm * 128 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * h) + i;
            if (ABL == 3 ? (row < Mm && acc[i][0][e] == 123.456f) : row < Mm) {
                float* const dp = D + (size_t)row * Nn + col;
                if (col + 1 < Nn) { const f32x2 v = {acc[i][0][e], acc[i][1][e]}; *reinterpret_cast<f32x2*>(dp) = v; }
                else if (col < Nn) dp[0] = acc[i][0][e];
            }
        }
}

// dw [Co][3][3][Ci] = G^T (sum of the f planes of P [NX][Co][Cip]) G; one thread = one (co, ci)
template <int MT>
__global__ __launch_bounds__(256) void k_w2d_wgrad_out(const float* __restrict__ P, float* __restrict__ dw, int Co, int Ci, int Cip, int f) {
    typedef W2T<MT> TR;
    constexpr int NT = TR::NT, NX = TR::NX;
    const size_t total = (size_t)Co * Ci, plane = (size_t)Co * Cip;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ci = (int)(idx % Ci);
        const size_t co = idx / Ci;
        const float* const pp = P + co * Cip + ci;
        // all NX values of a plane are loaded before they are added to the running sums: NX independent loads in flight per
        // plane (a per-element loop over the planes left one dependent load chain per thread: 0.28 of the HBM rate)
        float mx[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) mx[q] = pp[(size_t)q * plane];
        for (int k = 1; k < f; ++k) {
            float v[NX];
#pragma unroll
            for (int q = 0; q < NX; ++q) v[q] = pp[(size_t)(k * NX + q) * plane];
#pragma unroll
            for (int q = 0; q < NX; ++q) mx[q] += v[q];
        }
        float t[NT][3];         // rows a of P, transformed along b
#pragma unroll
        for (int a = 0; a < NT; ++a) TR::gt(mx + a * NT, t[a]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float col[NT], w3[3];
#pragma unroll
            for (int a = 0; a < NT; ++a) col[a] = t[a][kx];
            TR::gt(col, w3);
            dw[(co * 9 + 0 + kx) * Ci + ci] = w3[0];
            dw[(co * 9 + 3 + kx) * Ci + ci] = w3[1];
            dw[(co * 9 + 6 + kx) * Ci + ci] = w3[2];
        }
    }
}

// ---- output transform: M [NX][T][ldm] -> y [N,H,W,ldy] (+bias, + BatchNorm statistics partials with pixel counts) ----------------
// block = w2_tb(T) tiles x CH channels (CH = VW * cvn <= 256); thread = one channel vector, tiles pl apart
template <int MT, bool STATS>
__global__ __launch_bounds__(256) void k_w2d_output(const float* __restrict__ Mo, int ldm, const float* __restrict__ bias,
                                                   float* __restrict__ Y, int ldy, float* __restrict__ stats,
                                                   float* __restrict__ counts, int P, int H, int W, int th, int tw, int T,
                                                   int Cout, int cvn, int BM, int tmn, int tilesN, int split_start, int f) {
    typedef W2T<MT> TR;
    typedef typename TR::VT VT;
    constexpr int NT = TR::NT, NX = TR::NX, VW = TR::VW;
    __shared__ float red[2][256 * VW];
    const int t = threadIdx.x;
    const int pl = 256 / cvn, cv = t % cvn, lanep = t / cvn;
    const int c = (blockIdx.y * cvn + cv) * VW;
    const bool cok = c < Cout;                          // Cout % 4 == 0 (checked by the host)
    VT sh = {};
    if (cok && bias != nullptr) sh = *reinterpret_cast<const VT*>(bias + c);
    VT s1 = {}, s2 = {};
    const size_t plane = (size_t)T * ldm;
    const int TB = w2_tb(T);
    const int tile_end = min(T, (int)(blockIdx.x + 1) * TB);
    for (int tile = blockIdx.x * TB + lanep; tile < tile_end && cok; tile += pl) {
        const int n = tile / (th * tw), rr = tile - n * th * tw, ty = rr / tw, tx = rr - ty * tw;
        const float* const mp = Mo + (size_t)tile * ldm + c;
        // GEMM tile of plane xi: xi * tmn + (tile / BM) * tilesN + c / 128; ids >= split_start carry f K-range planes
        const int xi0 = f > 1 ? (split_start - ((tile / BM) * tilesN + c / W2_BN) + tmn - 1) / tmn : NX;
        // column j of the NT x NT product tile: A^T along the rows, then its contribution A^T[.][j] to the MT output columns:
        // MT x MT accumulators instead of an MT x NT array
        // (F(4x4) with float4: forcing <= 128 registers through launch bounds made hipcc spill: 4x slower; it hoists all loads at 215-230)
        VT o[MT][MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < MT; ++jj) o[i][jj] = VT{};
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            VT m[NT], z[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                m[i] = *reinterpret_cast<const VT*>(mp + (size_t)(i * NT + j) * plane);
                if (i * NT + j >= xi0)
                    for (int k = 1; k < f; ++k) m[i] += *reinterpret_cast<const VT*>(mp + (size_t)(k * NX + i * NT + j) * plane);
            }
            if constexpr (MT == 4) {
                w2_at(m, z);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (j == 0) { o[i][0] = z[i]; }
                    else if (j == 1) { o[i][0] += z[i]; o[i][1] = z[i]; o[i][2] = z[i]; o[i][3] = z[i]; }
                    else if (j == 2) { o[i][0] += z[i]; o[i][1] -= z[i]; o[i][2] += z[i]; o[i][3] -= z[i]; }
                    else if (j == 3) { o[i][0] += z[i]; o[i][1] += 2.f * z[i]; o[i][2] += 4.f * z[i]; o[i][3] += 8.f * z[i]; }
                    else if (j == 4) { o[i][0] += z[i]; o[i][1] -= 2.f * z[i]; o[i][2] += 4.f * z[i]; o[i][3] -= 8.f * z[i]; }
                    else { o[i][3] += z[i]; }
                }
            } else {
                w6_at(m, z);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int jj = 0; jj < MT; ++jj) {
                        if (TR::at(jj, j) != 0.f) o[i][jj] += TR::at(jj, j) * z[i];      // folds after unrolling
                    }
            }
        }
        const int yb = MT * ty, xb = MT * tx;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                if (yb + i < H && xb + j < W) {
                    *reinterpret_cast<VT*>(Y + ((size_t)(n * H + yb + i) * W + xb + j) * ldy + c) = o[i][j] + sh;
                    if (STATS) { s1 += o[i][j]; s2 += o[i][j] * o[i][j]; }
                }
            }
        }
    }
    if (!STATS) return;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
        red[0][t * VW + j] = s1[j];
        red[1][t * VW + j] = s2[j];
    }
    __syncthreads();
    if (t < cvn && c < Cout) {
        int cnt = 0;
        for (int tile = blockIdx.x * TB; tile < tile_end; ++tile) {
            const int rr = tile % (th * tw), ty = rr / tw, tx = rr - ty * tw;
            cnt += min(MT, H - MT * ty) * min(MT, W - MT * tx);
        }
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < pl; ++p) {
                a += red[0][(p * cvn + t) * VW + j];
                b += red[1][(p * cvn + t) * VW + j];
            }
            const float m2 = b - a * a / (float)cnt;     // about the partial mean; sums exclude the bias (shift invariance)
            stats[(size_t)blockIdx.x * Cout + c + j] = a + (float)cnt * sh[j];
            stats[(size_t)(P + blockIdx.x) * Cout + c + j] = m2 > 0.f ? m2 : 0.f;
        }
        if (t == 0 && blockIdx.y == 0) counts[blockIdx.x] = (float)cnt;
    }
}

inline int w2_tiles(int mt, int N, int H, int W) { return N * ((H + mt - 1) / mt) * ((W + mt - 1) / mt); }
inline int w2_nx(int mt) { return (mt + 2) * (mt + 2); }

}  // namespace

// How the NX GEMMs are cut into workgroups: tile height, grid, and the K split of the last partial round.
struct W2Plan { int BM, tilesM, tilesN, NT, split_start, f; };
static W2Plan plan_w2d(int nx, int T, int Cin, int Cout) {
    W2Plan p;
    p.BM = 128;
    p.tilesM = cvk_cdiv(T, p.BM);
    p.tilesN = cvk_cdiv(Cout, W2_BN);
    p.NT = nx * p.tilesM * p.tilesN;
    const int slots = 512;                                // resident workgroups: two per CU
    const int nK = Cin / 32;
    const int full = p.NT / slots * slots, R = p.NT - full;
    p.f = 1;
    if (R > 0) {
        p.f = slots / R < 4 ? slots / R : 4;
        if (p.f > nK / 4) p.f = nK / 4;
        if (p.f < 1) p.f = 1;
    }
    p.split_start = p.f > 1 ? full : p.NT;
    return p;
}

static inline int w2_tpad(int T) { return cvk_cdiv(T, 32) * 32; }

extern "C" int cvk_w2d_tpad(int T) { return T > 0 ? w2_tpad(T) : 0; }

// ---- implementations, mt = 4: F(4x4,3x3) (cvk_w2d_*), mt = 6: F(6x6,3x3) (cvk_w6_*) ------------------------------------------
static int w2i_tiles(int mt, int N, int H, int W) { return (N > 0 && H > 0 && W > 0) ? w2_tiles(mt, N, H, W) : 0; }

static int w2i_stat_partials(int mt, int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int T = w2_tiles(mt, N, H, W);
    return cvk_cdiv(T, w2_tb(T));
}

static int w2i_ksplit(int mt, int T, int Cin, int Cout) {
    if (T <= 0 || Cin < 32 || Cout <= 0) return 0;
    return plan_w2d(w2_nx(mt), T, Cin, Cout).f;
}

static size_t w2i_workspace_bytes(int mt, int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 32 || Cout <= 0) return 0;
    const int T = w2_tiles(mt, N, H, W), nx = w2_nx(mt);
    return ((size_t)nx * w2_tpad(T) * Cin + 128 + (size_t)nx * T * plan_w2d(nx, T, Cin, Cout).f * Cout) * sizeof(float);
}

static int w2i_weight_transform(int mt, const char* who, const float* w, float* U, int Cout, int Cin, void* stream) {
    CVK_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "%s: bad arguments", who);
    const size_t total = (size_t)Cout * Cin;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (mt == 4) hipLaunchKernelGGL(k_w2d_weight<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin, (const unsigned*)nullptr, CvkSplitTab{});
    else hipLaunchKernelGGL(k_w2d_weight<6>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin, (const unsigned*)nullptr, CvkSplitTab{});
    CVK_LAUNCH_RETURN(who);
}

static int w2i_weight_transform_dgrad(int mt, const char* who, const float* w, float* U, int Cout, int Cin, void* stream) {
    CVK_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "%s: bad arguments", who);
    const dim3 grid(cvk_cdiv(Cin, 32), cvk_cdiv(Cout, 32));
    if (mt == 4) hipLaunchKernelGGL(k_w2d_weight_dgrad<4>, grid, dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin, (const unsigned*)nullptr, CvkSplitTab{});
    else hipLaunchKernelGGL(k_w2d_weight_dgrad<6>, grid, dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin, (const unsigned*)nullptr, CvkSplitTab{});
    CVK_LAUNCH_RETURN(who);
}

extern "C" int cvk_w2d_weight_transform_batch(const cvk_wt_job* jobs, int n, void* stream) {
    CVK_CHECK_ARG(jobs && n > 0 && n <= CVK_WT_BATCH_MAX, "cvk_w2d_weight_transform_batch: 1..%d jobs", CVK_WT_BATCH_MAX);
    W2WJobsDev d;
    unsigned nb = 0;
    for (int i = 0; i < n; ++i) {
        const cvk_wt_job& q = jobs[i];
        CVK_CHECK_ARG(q.w && q.out && q.rows > 0 && q.cols > 0 && (q.tile == 4 || q.tile == 6), "cvk_w2d_weight_transform_batch: bad job %d", i);
        d.w[i] = q.w; d.out[i] = q.out; d.Co[i] = q.rows; d.Ci[i] = q.cols; d.kind[i] = q.tile + (q.dgrad ? 16 : 0);
        d.first[i] = nb;
        if (q.dgrad) nb += (unsigned)(cvk_cdiv(q.cols, 32) * cvk_cdiv(q.rows, 32));
        else { const size_t total = (size_t)q.rows * q.cols; nb += (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096); }
    }
    d.first[n] = nb; d.n = n;
    hipLaunchKernelGGL(k_w2d_weight_batch, dim3(nb), dim3(256), 0, (hipStream_t)stream, d);
    CVK_LAUNCH_RETURN("cvk_w2d_weight_transform_batch");
}

static int w2i_input_transform(int mt, const char* who, const float* x, float* V, int N, int H, int W, int Cin, void* stream) {
    CVK_CHECK_ARG(x && V && N > 0 && H > 0 && W > 0 && Cin >= 4 && Cin % 4 == 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(V), "%s: pointers must be 16-byte aligned", who);
    const int th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw, Tpad = w2_tpad(T);
    const long threads = (long)Tpad * (Cin / (mt == 4 ? 4 : 2));
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (mt == 4) hipLaunchKernelGGL(k_w2d_input<4>, grid, dim3(256), 0, (hipStream_t)stream, x, V, H, W, Cin, th, tw, T, Tpad, (const unsigned*)nullptr, CvkSplitTab{});
    else hipLaunchKernelGGL(k_w2d_input<6>, grid, dim3(256), 0, (hipStream_t)stream, x, V, H, W, Cin, th, tw, T, Tpad, (const unsigned*)nullptr, CvkSplitTab{});
    CVK_LAUNCH_RETURN(who);
}

static int w2i_gemm(int mt, const char* who, const float* V, const float* U, float* Mo, int T, int Cin, int Cout, void* stream) {
    CVK_CHECK_ARG(V && U && Mo && T > 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(Cin >= 32 && Cin % 32 == 0 && Cout > 0, "%s: Cin=%d must be a multiple of 32", who, Cin);
    CVK_CHECK_ARG(cvk_aligned16(V) && cvk_aligned16(U) && cvk_aligned16(Mo), "%s: pointers must be 16-byte aligned", who);
    const int nx = w2_nx(mt);
    CVK_CHECK_ARG((long)T * nx * (Cin > Cout ? Cin : Cout) < (1L << 40), "%s: tensor too large", who);
    const W2Plan p = plan_w2d(nx, T, Cin, Cout);
    const dim3 grid(p.split_start + (p.NT - p.split_start) * p.f);
    const size_t part_stride = (size_t)nx * T * Cout;
    hipStream_t s = (hipStream_t)stream;
    static int cus = 0;          // queried once (benign race: every thread stores the same value)
    if (cus == 0) {
        int dev = 0, v = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    const int no_stagger = cvk_knob("CVK_W2D_NO_STAGGER", 0);          // experiments build: A/B of the staggered start
    const int stagger = (!no_stagger && (int)grid.x >= 2 * cus && p.split_start >= 2 * cus) ? cus : 0;
#ifdef CVK_EXPERIMENTS
#define CVK_W2D_ABL_GO(A_) hipLaunchKernelGGL((k_w2d_gemm<128, 32, 2, 2, A_>), grid, dim3(256), 0, s, V, U, Mo, T, w2_tpad(T), Cout, Cin, Cout, p.tilesM, p.tilesN, p.split_start, p.f, part_stride, stagger)
    switch (cvk_knob("CVK_W2D_ABL", 0)) {           // ablation variants (WRONG results): where the GEMM's time goes
        case 1: CVK_W2D_ABL_GO(1); CVK_LAUNCH_RETURN(who);
        case 2: CVK_W2D_ABL_GO(2); CVK_LAUNCH_RETURN(who);
        case 3: CVK_W2D_ABL_GO(3); CVK_LAUNCH_RETURN(who);
        case 4: CVK_W2D_ABL_GO(4); CVK_LAUNCH_RETURN(who);
        case 5: CVK_W2D_ABL_GO(5); CVK_LAUNCH_RETURN(who);
        case 6: CVK_W2D_ABL_GO(6); CVK_LAUNCH_RETURN(who);
        case 8: CVK_W2D_ABL_GO(8); CVK_LAUNCH_RETURN(who);
        case 10: CVK_W2D_ABL_GO(10); CVK_LAUNCH_RETURN(who);
        case 11: CVK_W2D_ABL_GO(11); CVK_LAUNCH_RETURN(who);
        case 12: CVK_W2D_ABL_GO(12); CVK_LAUNCH_RETURN(who);
        default: break;
    }
#endif
    hipLaunchKernelGGL((k_w2d_gemm<128, 32, 2, 2>), grid, dim3(256), 0, s, V, U, Mo, T, w2_tpad(T), Cout, Cin, Cout, p.tilesM, p.tilesN,
                       p.split_start, p.f, part_stride, stagger);
    CVK_LAUNCH_RETURN(who);
}

static int w2i_output(int mt, const char* who, const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H,
                      int W, int Cin, int Cout, int ldy, void* stream) {
    CVK_CHECK_ARG(Mo && y && N > 0 && H > 0 && W > 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(Cout >= 64 && Cout % 4 == 0 && ldy >= Cout && ldy % 4 == 0, "%s: needs Cout %% 4 == 0, Cout >= 64 (got %d)", who, Cout);
    CVK_CHECK_ARG((stats == nullptr) == (counts == nullptr), "%s: stats and counts go together", who);
    CVK_CHECK_ARG(cvk_aligned16(Mo) && cvk_aligned16(y), "%s: pointers must be 16-byte aligned", who);
    CVK_CHECK_ARG(Cin >= 32 && Cin % 32 == 0, "%s: Cin (the GEMM depth, which fixes the K split of the planes) must be a multiple of 32", who);
    const int th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw;
    const int vw = mt == 4 ? 4 : 2, nv = Cout / vw;
    const int cvn = nv >= 64 ? 64 : (nv >= 32 ? 32 : 16);
    const int P = cvk_cdiv(T, w2_tb(T));
    const W2Plan p = plan_w2d(w2_nx(mt), T, Cin, Cout);
    dim3 grid(P, cvk_cdiv(nv, cvn));
    hipStream_t s = (hipStream_t)stream;
#define CVK_W2_OUT(MT_, ST_) hipLaunchKernelGGL((k_w2d_output<MT_, ST_>), grid, dim3(256), 0, s, Mo, Cout, bias, y, ldy, stats, counts, P, H, W, th, tw, T, Cout, cvn, \
                                                p.BM, p.tilesM * p.tilesN, p.tilesN, p.split_start, p.f)
    if (mt == 4) { if (stats) CVK_W2_OUT(4, true); else CVK_W2_OUT(4, false); }
    else         { if (stats) CVK_W2_OUT(6, true); else CVK_W2_OUT(6, false); }
#undef CVK_W2_OUT
    CVK_LAUNCH_RETURN(who);
}

// ---- weight-grad ---------------------------------------------------------------------------------------------------------
struct W2WPlan { int Tpad, tilesM, tilesN, f; };
static W2WPlan plan_w2d_wgrad(int nx, int T, int Cin_pad, int Cout) {
    W2WPlan p;
    p.Tpad = w2_tpad(T);
    p.tilesM = cvk_cdiv(Cout, 128);
    p.tilesN = cvk_cdiv(Cin_pad, 128);
    const int nt = nx * p.tilesM * p.tilesN, nK = p.Tpad / 32;
    // depth split: fill the 512 resident workgroups at least twice over, keep >= 4 slices per range, at most 16 planes
    int f = cvk_cdiv(1024, nt);
    if (f > nK / 4) f = nK / 4;
    if (f > 16) f = 16;
    if (f < 1) f = 1;
    p.f = f;
    return p;
}

static int w2i_wgrad_ksplit(int mt, int T, int Cin_pad, int Cout) {
    if (T <= 0 || Cin_pad <= 0 || Cout <= 0) return 0;
    return plan_w2d_wgrad(w2_nx(mt), T, Cin_pad, Cout).f;
}

static int w2i_dy_transform(int mt, const char* who, const float* dy, int ld_dy, float* E, int N, int H, int W, int Cout, void* stream) {
    CVK_CHECK_ARG(dy && E && N > 0 && H > 0 && W > 0 && Cout >= 4 && Cout % 4 == 0 && ld_dy >= Cout && ld_dy % 4 == 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(cvk_aligned16(dy) && cvk_aligned16(E), "%s: pointers must be 16-byte aligned", who);
    const int th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw, Tpad = w2_tpad(T);
    const long threads = (long)Tpad * (Cout / (mt == 4 ? 4 : 2));
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (mt == 4) hipLaunchKernelGGL(k_w2d_dy<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, ld_dy, E, H, W, Cout, th, tw, T, Tpad);
    else hipLaunchKernelGGL(k_w2d_dy<6>, grid, dim3(256), 0, (hipStream_t)stream, dy, ld_dy, E, H, W, Cout, th, tw, T, Tpad);
    CVK_LAUNCH_RETURN(who);
}

static int w2i_gemm_tn(int mt, const char* who, const float* E, const float* V, float* P, int T, int Cin_pad, int Cout, void* stream) {
    CVK_CHECK_ARG(E && V && P && T > 0 && Cin_pad > 0 && Cin_pad % 4 == 0 && Cout > 0 && Cout % 4 == 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(cvk_aligned16(E) && cvk_aligned16(V) && cvk_aligned16(P), "%s: pointers must be 16-byte aligned", who);
    const int nx = w2_nx(mt);
    const W2WPlan p = plan_w2d_wgrad(nx, T, Cin_pad, Cout);
#ifdef CVK_EXPERIMENTS
#define CVK_TN_ABL_GO(A_) hipLaunchKernelGGL(k_w2d_gemm_tn<A_>, dim3(nx * p.tilesM * p.tilesN * p.f), dim3(256), 0, (hipStream_t)stream, E, Cout, V, Cin_pad, P, p.Tpad, Cout, Cin_pad, p.tilesM, p.tilesN, p.f, nx)
    switch (cvk_knob("CVK_W2D_TN_ABL", 0)) {
        case 1: CVK_TN_ABL_GO(1); CVK_LAUNCH_RETURN(who);
        case 2: CVK_TN_ABL_GO(2); CVK_LAUNCH_RETURN(who);
        case 3: CVK_TN_ABL_GO(3); CVK_LAUNCH_RETURN(who);
        case 4: CVK_TN_ABL_GO(4); CVK_LAUNCH_RETURN(who);
        case 5: CVK_TN_ABL_GO(5); CVK_LAUNCH_RETURN(who);
        default: break;
    }
#endif
    hipLaunchKernelGGL(k_w2d_gemm_tn<0>, dim3(nx * p.tilesM * p.tilesN * p.f), dim3(256), 0, (hipStream_t)stream, E, Cout, V, Cin_pad, P, p.Tpad,
                       Cout, Cin_pad, p.tilesM, p.tilesN, p.f, nx);
    CVK_LAUNCH_RETURN(who);
}

static int w2i_wgrad_output(int mt, const char* who, const float* P, float* dw, int T, int Cin, int Cin_pad, int Cout, void* stream) {
    CVK_CHECK_ARG(P && dw && T > 0 && Cin > 0 && Cin <= Cin_pad && Cout > 0, "%s: bad arguments", who);
    const W2WPlan p = plan_w2d_wgrad(w2_nx(mt), T, Cin_pad, Cout);
    const size_t total = (size_t)Cout * Cin;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (mt == 4) hipLaunchKernelGGL(k_w2d_wgrad_out<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, dw, Cout, Cin, Cin_pad, p.f);
    else hipLaunchKernelGGL(k_w2d_wgrad_out<6>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, dw, Cout, Cin, Cin_pad, p.f);
    CVK_LAUNCH_RETURN(who);
}

static size_t w2i_wgrad_workspace_bytes(int mt, int N, int H, int W, int Cin_pad, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_pad <= 0 || Cout <= 0) return 0;
    const int nx = w2_nx(mt);
    const W2WPlan p = plan_w2d_wgrad(nx, w2_tiles(mt, N, H, W), Cin_pad, Cout);
    // V and E planes (+ 512 bytes of slack each: partial column tiles read past the last row) and the f product planes
    return ((size_t)nx * p.Tpad * ((size_t)Cin_pad + Cout) + 2 * 128 + (size_t)p.f * nx * Cout * Cin_pad) * sizeof(float);
}

static int w2i_dy_both(int mt, const char* who, const float* dy, int ld_dy, float* Vp, float* E, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dy && Vp && E && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && ld_dy >= C && ld_dy % 4 == 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(cvk_aligned16(dy) && cvk_aligned16(Vp) && cvk_aligned16(E), "%s: pointers must be 16-byte aligned", who);
    const int th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw, Tpad = w2_tpad(T);
    const long threads = (long)Tpad * (C / (mt == 4 ? 4 : 2));
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (mt == 4) hipLaunchKernelGGL(k_w2d_dy_both<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, ld_dy, Vp, E, H, W, C, th, tw, T, Tpad, Tpad, (const unsigned*)nullptr, CvkSplitTab{}, CvkSplitTab{});
    else hipLaunchKernelGGL(k_w2d_dy_both<6>, grid, dim3(256), 0, (hipStream_t)stream, dy, ld_dy, Vp, E, H, W, C, th, tw, T, Tpad, Tpad, (const unsigned*)nullptr, CvkSplitTab{}, CvkSplitTab{});
    CVK_LAUNCH_RETURN(who);
}

// ---- split-operand path (opt-in; csrc/split3.hip holds the GEMMs, split_fmt.h the formats): transforms that write split planes ------------
// exponent tables from the transform routines themselves: c[i] = ceil(log2(sum_k |M[i][k]|)), M applied to the unit vectors
template <int MT>
static CvkSplitTab w2_split_tab(int kind) {
    typedef W2T<MT> TR;
    constexpr int NT = TR::NT;
    const int nin = kind == CVK_SPLIT_KIND_B ? NT : (kind == CVK_SPLIT_KIND_G ? 3 : MT);
    double rs[8] = {};
    for (int k = 0; k < nin; ++k) {
        float d[8] = {}, v[8] = {};
        d[k] = 1.f;
        if (kind == CVK_SPLIT_KIND_B) TR::bt(d, v);
        else if (kind == CVK_SPLIT_KIND_G) TR::g(d, v);
        else TR::a(d, v);
        for (int i = 0; i < NT; ++i) rs[i] += v[i] < 0.f ? -(double)v[i] : (double)v[i];
    }
    CvkSplitTab t;
    t.c8 = 0ull;
    t.nt = NT;
    for (int i = 0; i < NT; ++i) {
        int c = -20;
        while (rs[i] > ldexp(1.0, c) * (1.0 + 1e-6)) ++c;        // 2^c >= the row sum (fp32 coefficient rounding allowed for)
        t.c8 |= (unsigned long long)(unsigned)(c + 64) << (8 * i);
    }
    return t;
}
CvkSplitTab cvk_split_tab(int tile, int kind) { return tile == 4 ? w2_split_tab<4>(kind) : w2_split_tab<6>(kind); }

// exponent e of the scale 2^e of transform index xi of a fmt-2 plane (kind 0 = input transforms B, 1 = filter transforms G, 2 = dy -> E A) for a
// source tensor whose largest magnitude has the bit pattern amax_bits: what the transforms apply and the GEMM epilogues undo (tests)
extern "C" int cvk_split_scale_exponent(int tile, int kind, int xi, unsigned amax_bits) {
    if (!(tile == 4 || tile == 6) || kind < 0 || kind > 2 || xi < 0 || xi >= w2_nx(tile)) return 0;
    const CvkSplitTab t = cvk_split_tab(tile, kind);
    return cvk_split_exp(amax_bits, cvk_split_tab_c(t, xi / t.nt), cvk_split_tab_c(t, xi % t.nt));
}

// largest magnitude of x [rows][C] (row stride ld) as an fp32 bit pattern, combined into the amax block (cvk_common.h) by atomicMax: the caller
// zeroes the block (cvk_amax_block_words() words).
// Magnitudes order like their bit patterns, so the maximum is exact and independent of the order (bitwise reproducible).  Four 16-byte loads
// per thread in flight; a dense tensor (ld == C) is walked as one flat array.
typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned absmax4(u32x4a v, unsigned m) {
    v &= 0x7FFFFFFFu;
    const unsigned m01 = v[0] > v[1] ? v[0] : v[1], m23 = v[2] > v[3] ? v[2] : v[3];
    const unsigned mv = m01 > m23 ? m01 : m23;
    return mv > m ? mv : m;
}
template <bool DENSE>
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ X, long rows, int C, int ld, unsigned* __restrict__ out) {
    const int c4n = C >> 2;
    const long total = rows * c4n;                          // 16-byte pieces
    const long step = (long)gridDim.x * 256;
    unsigned m = 0u;
    auto piece = [&](long i) -> const u32x4a* {
        if (DENSE) return reinterpret_cast<const u32x4a*>(X) + i;
        const long r = i / c4n;
        return reinterpret_cast<const u32x4a*>(X + r * ld + ((int)(i - r * c4n) << 2));
    };
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < total; i += 4 * step) {
        const u32x4a v0 = *piece(i), v1 = *piece(i + step), v2 = *piece(i + 2 * step), v3 = *piece(i + 3 * step);
        m = absmax4(v0, m); m = absmax4(v1, m); m = absmax4(v2, m); m = absmax4(v3, m);
    }
    for (; i < total; i += step) m = absmax4(*piece(i), m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o);
        m = other > m ? other : m;
    }
    // one atomic per workgroup, into the slot of its number, and only when it can raise it (cvk_common.h, amax blocks)
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a01 = wm[0] > wm[1] ? wm[0] : wm[1], a23 = wm[2] > wm[3] ? wm[2] : wm[3];
        const unsigned bm = a01 > a23 ? a01 : a23;
        unsigned* const slot = out + (size_t)(blockIdx.x % CVK_AMAX_SLOTS) * CVK_AMAX_STRIDE;
        if (bm > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, bm);
    }
}
extern "C" int cvk_amax_block_words(void) { return CVK_AMAX_WORDS; }
// the value of an amax block copied to the host (n = cvk_amax_block_words() words): the maximum over its slots
extern "C" unsigned cvk_amax_block_value(const unsigned* host_words, int n) {
    unsigned m = 0u;
    for (int i = 0; i + 1 <= n; i += CVK_AMAX_STRIDE) m = host_words[i] > m ? host_words[i] : m;
    return m;
}
extern "C" int cvk_absmax_f32(const float* x, long rows, int C, int ld, void* amax_bits, void* stream) {
    CVK_CHECK_ARG(x && amax_bits && rows > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0, "cvk_absmax_f32: bad arguments (C, ld multiples of 4)");
    CVK_CHECK_ARG(cvk_aligned16(x), "cvk_absmax_f32: x must be 16-byte aligned");
    const long total = rows * (C / 4);
    const long want = (total + 1023) / 1024;                // four pieces per thread and round
    const int blocks = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
    if (ld == C) hipLaunchKernelGGL(k_absmax<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, (unsigned*)amax_bits);
    else hipLaunchKernelGGL(k_absmax<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, (unsigned*)amax_bits);
    CVK_LAUNCH_RETURN("cvk_absmax_f32");
}

#define CVK_SPLIT_FMT_CHECK(who) \
    CVK_CHECK_ARG(fmt == 3 || (fmt == 2 && amax != nullptr), "%s: fmt is 3 (bf16 x 3) or 2 (fp16 x 2, needs the source tensor's cvk_absmax_f32 word)", who)

// x -> V as split planes (rows padded to 256); amax: the cvk_absmax_f32 word of x (fmt 2)
extern "C" int cvk_w2d_input_transform_split(int fmt, int tile, const float* x, void* V, const void* amax, int N, int H, int W, int Cin, void* stream) {
    const char* who = "cvk_w2d_input_transform_split";
    CVK_CHECK_ARG((tile == 4 || tile == 6) && x && V && N > 0 && H > 0 && W > 0 && Cin >= 32 && Cin % 32 == 0, "%s: bad arguments", who);
    CVK_SPLIT_FMT_CHECK(who);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(V), "%s: pointers must be 16-byte aligned", who);
    const int mt = tile, th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw, Tpad = cvk_split3_rows_pad(T, 256);
    const long threads = (long)Tpad * (Cin / (mt == 4 ? 4 : 2));
    const dim3 grid((unsigned)((threads + 255) / 256));
    const CvkSplitTab tab = cvk_split_tab(tile, CVK_SPLIT_KIND_B);
    const unsigned* am = (const unsigned*)amax;
#define CVK_W2_INS(MT_, F_) hipLaunchKernelGGL((k_w2d_input<MT_, F_>), grid, dim3(256), 0, (hipStream_t)stream, x, (float*)V, H, W, Cin, th, tw, T, Tpad, am, tab)
    if (mt == 4) { if (fmt == 3) CVK_W2_INS(4, 3); else CVK_W2_INS(4, 2); }
    else         { if (fmt == 3) CVK_W2_INS(6, 3); else CVK_W2_INS(6, 2); }
#undef CVK_W2_INS
    CVK_LAUNCH_RETURN(who);
}
extern "C" int cvk_w2d_input_transform_split3(int tile, const float* x, void* V3, int N, int H, int W, int Cin, void* stream) {
    return cvk_w2d_input_transform_split(3, tile, x, V3, nullptr, N, H, W, Cin, stream);
}

// dy -> V' as split planes (rows padded to 256) and E: fp32 planes [NX][cvk_w2d_tpad(T)][C] (e_split = 0) or split planes (rows padded to 256);
// amax: the cvk_absmax_f32 word of dy (fmt 2; V' is scaled with the B table, E with the A table)
extern "C" int cvk_w2d_dy_transform_both_split(int fmt, int tile, const float* dy, int ld_dy, void* Vp, void* E, int e_split, const void* amax,
                                               int N, int H, int W, int C, void* stream) {
    const char* who = "cvk_w2d_dy_transform_both_split";
    CVK_CHECK_ARG((tile == 4 || tile == 6) && dy && Vp && E && N > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0 && ld_dy >= C && ld_dy % 4 == 0,
                  "%s: bad arguments", who);
    CVK_SPLIT_FMT_CHECK(who);
    CVK_CHECK_ARG(cvk_aligned16(dy) && cvk_aligned16(Vp) && cvk_aligned16(E), "%s: pointers must be 16-byte aligned", who);
    const int mt = tile, th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw, Tpad = cvk_split3_rows_pad(T, 256);
    const int TpadE = e_split ? Tpad : w2_tpad(T);
    const long threads = (long)Tpad * (C / (mt == 4 ? 4 : 2));
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    const CvkSplitTab tB = cvk_split_tab(tile, CVK_SPLIT_KIND_B), tA = cvk_split_tab(tile, CVK_SPLIT_KIND_A);
    const unsigned* am = (const unsigned*)amax;
#define CVK_W2_DYS(MT_, S_, F_) hipLaunchKernelGGL((k_w2d_dy_both<MT_, S_, F_>), grid, dim3(256), 0, s, dy, ld_dy, (float*)Vp, (float*)E, H, W, C, th, tw, T, Tpad, TpadE, am, tB, tA)
    if (mt == 4) {
        if (fmt == 3) { if (e_split) CVK_W2_DYS(4, 3, 3); else CVK_W2_DYS(4, 1, 3); }
        else          { if (e_split) CVK_W2_DYS(4, 3, 2); else CVK_W2_DYS(4, 1, 2); }
    } else {
        if (fmt == 3) { if (e_split) CVK_W2_DYS(6, 3, 3); else CVK_W2_DYS(6, 1, 3); }
        else          { if (e_split) CVK_W2_DYS(6, 3, 2); else CVK_W2_DYS(6, 1, 2); }
    }
#undef CVK_W2_DYS
    CVK_LAUNCH_RETURN(who);
}
extern "C" int cvk_w2d_dy_transform_both_split3(int tile, const float* dy, int ld_dy, void* Vp3, void* E, int e_split, int N, int H, int W, int C,
                                                void* stream) {
    return cvk_w2d_dy_transform_both_split(3, tile, dy, ld_dy, Vp3, E, e_split, nullptr, N, H, W, C, stream);
}

// filter -> U as split planes, written by the transform kernels' own store loops; dgrad: the rotated / channel-exchanged filter of the
// data-grad (rows = Cin, depth = Cout of the forward layer); amax: the cvk_absmax_f32 word of w (fmt 2)
extern "C" int cvk_w2d_weight_transform_split(int fmt, int tile, const float* w, void* U, const void* amax, int Cout, int Cin, int dgrad, void* stream) {
    const char* who = "cvk_w2d_weight_transform_split";
    CVK_CHECK_ARG((tile == 4 || tile == 6) && w && U && Cout > 0 && Cin > 0, "%s: bad arguments", who);
    CVK_SPLIT_FMT_CHECK(who);
    const int rows = dgrad ? Cin : Cout, cols = dgrad ? Cout : Cin;
    CVK_CHECK_ARG(cols % 32 == 0, "%s: the GEMM depth (%d) must be a multiple of 32", who, cols);
    hipStream_t s = (hipStream_t)stream;
    const int Rp = cvk_split3_rows_pad(rows, 128);
    if (Rp != rows) {       // padding rows are never written by the transform: zero planes first
        const hipError_t e = hipMemsetAsync(U, 0, (size_t)w2_nx(tile) * (cols / 32) * fmt * Rp * 64, s);
        if (e != hipSuccess) { cvk_set_error("%s: memset failed: %s", who, hipGetErrorString(e)); return (int)e; }
    }
    const CvkSplitTab tab = cvk_split_tab(tile, CVK_SPLIT_KIND_G);
    const unsigned* am = (const unsigned*)amax;
    if (dgrad) {
        const dim3 grid(cvk_cdiv(Cin, 32), cvk_cdiv(Cout, 32));
#define CVK_W2_WDS(MT_, F_) hipLaunchKernelGGL((k_w2d_weight_dgrad<MT_, F_>), grid, dim3(256), 0, s, w, (float*)U, Cout, Cin, am, tab)
        if (tile == 4) { if (fmt == 3) CVK_W2_WDS(4, 3); else CVK_W2_WDS(4, 2); }
        else           { if (fmt == 3) CVK_W2_WDS(6, 3); else CVK_W2_WDS(6, 2); }
#undef CVK_W2_WDS
    } else {
        const size_t total = (size_t)Cout * Cin;
        const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
#define CVK_W2_WS(MT_, F_) hipLaunchKernelGGL((k_w2d_weight<MT_, F_>), dim3(blocks), dim3(256), 0, s, w, (float*)U, Cout, Cin, am, tab)
        if (tile == 4) { if (fmt == 3) CVK_W2_WS(4, 3); else CVK_W2_WS(4, 2); }
        else           { if (fmt == 3) CVK_W2_WS(6, 3); else CVK_W2_WS(6, 2); }
#undef CVK_W2_WS
    }
    CVK_LAUNCH_RETURN(who);
}
extern "C" int cvk_w2d_weight_transform_split3(int tile, const float* w, void* U3, float* tmp, int Cout, int Cin, int dgrad, void* stream) {
    (void)tmp;              // unused since the fused version; kept in the signature
    return cvk_w2d_weight_transform_split(3, tile, w, U3, nullptr, Cout, Cin, dgrad, stream);
}

// the output pass for product planes WITHOUT K-range partials (Mo [NX][T][Cout], one plane per transform index)
extern "C" int cvk_w2d_output_plain(int tile, const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H, int W,
                                    int Cout, int ldy, void* stream) {
    const char* who = "cvk_w2d_output_plain";
    CVK_CHECK_ARG((tile == 4 || tile == 6) && Mo && y && N > 0 && H > 0 && W > 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(Cout >= 64 && Cout % 4 == 0 && ldy >= Cout && ldy % 4 == 0, "%s: needs Cout %% 4 == 0, Cout >= 64 (got %d)", who, Cout);
    CVK_CHECK_ARG((stats == nullptr) == (counts == nullptr), "%s: stats and counts go together", who);
    CVK_CHECK_ARG(cvk_aligned16(Mo) && cvk_aligned16(y), "%s: pointers must be 16-byte aligned", who);
    const int mt = tile, th = (H + mt - 1) / mt, tw = (W + mt - 1) / mt, T = N * th * tw;
    const int vw = mt == 4 ? 4 : 2, nv = Cout / vw;
    const int cvn = nv >= 64 ? 64 : (nv >= 32 ? 32 : 16);
    const int P = cvk_cdiv(T, w2_tb(T));
    const int tilesM = cvk_cdiv(T, 128), tilesN = cvk_cdiv(Cout, 128);
    dim3 grid(P, cvk_cdiv(nv, cvn));
    hipStream_t s = (hipStream_t)stream;
#define CVK_W2_OUTP(MT_, ST_) hipLaunchKernelGGL((k_w2d_output<MT_, ST_>), grid, dim3(256), 0, s, Mo, Cout, bias, y, ldy, stats, counts, P, H, W, th, tw, T, Cout, cvn, \
                                                 128, tilesM * tilesN, tilesN, tilesM * tilesN, 1)
    if (mt == 4) { if (stats) CVK_W2_OUTP(4, true); else CVK_W2_OUTP(4, false); }
    else         { if (stats) CVK_W2_OUTP(6, true); else CVK_W2_OUTP(6, false); }
#undef CVK_W2_OUTP
    CVK_LAUNCH_RETURN(who);
}

// dW from product planes P [f][NX][Cout][Cin_pad] with an explicit number of K-range planes (the split weight-grad GEMM plans its own)
extern "C" int cvk_w2d_wgrad_output_f(int tile, const float* P, float* dw, int Cin, int Cin_pad, int Cout, int f, void* stream) {
    CVK_CHECK_ARG((tile == 4 || tile == 6) && P && dw && Cin > 0 && Cin <= Cin_pad && Cout > 0 && f >= 1 && f <= 64, "cvk_w2d_wgrad_output_f: bad arguments");
    const size_t total = (size_t)Cout * Cin;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (tile == 4) hipLaunchKernelGGL(k_w2d_wgrad_out<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, dw, Cout, Cin, Cin_pad, f);
    else hipLaunchKernelGGL(k_w2d_wgrad_out<6>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, dw, Cout, Cin, Cin_pad, f);
    CVK_LAUNCH_RETURN("cvk_w2d_wgrad_output_f");
}

// ---- C ABI: F(4x4,3x3) ----------------------------------------------------------------------------------------------------
extern "C" int cvk_w2d_tiles(int N, int H, int W) { return w2i_tiles(4, N, H, W); }
extern "C" int cvk_w2d_stat_partials(int N, int H, int W) { return w2i_stat_partials(4, N, H, W); }
extern "C" int cvk_w2d_ksplit(int T, int Cin, int Cout) { return w2i_ksplit(4, T, Cin, Cout); }
extern "C" size_t cvk_conv3x3_w2d_workspace_bytes(int N, int H, int W, int Cin, int Cout) { return w2i_workspace_bytes(4, N, H, W, Cin, Cout); }
extern "C" int cvk_w2d_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream) {
    return w2i_weight_transform(4, "cvk_w2d_weight_transform", w, U, Cout, Cin, stream);
}
extern "C" int cvk_w2d_weight_transform_dgrad(const float* w, float* U, int Cout, int Cin, void* stream) {
    return w2i_weight_transform_dgrad(4, "cvk_w2d_weight_transform_dgrad", w, U, Cout, Cin, stream);
}
extern "C" int cvk_w2d_input_transform(const float* x, float* V, int N, int H, int W, int Cin, void* stream) {
    return w2i_input_transform(4, "cvk_w2d_input_transform", x, V, N, H, W, Cin, stream);
}
extern "C" int cvk_w2d_gemm(const float* V, const float* U, float* Mo, int T, int Cin, int Cout, void* stream) {
    return w2i_gemm(4, "cvk_w2d_gemm", V, U, Mo, T, Cin, Cout, stream);
}
extern "C" int cvk_w2d_output(const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H, int W,
                              int Cin, int Cout, int ldy, void* stream) {
    return w2i_output(4, "cvk_w2d_output", Mo, bias, y, stats, counts, N, H, W, Cin, Cout, ldy, stream);
}

extern "C" int cvk_conv3x3_w2d(const float* x, const float* U, const float* bias, float* y, float* stats, float* counts,
                               int N, int H, int W, int Cin, int Cout, int ldy, void* workspace, size_t workspace_bytes,
                               void* stream) {
    CVK_CHECK_ARG(workspace && N > 0 && H > 0 && W > 0, "cvk_conv3x3_w2d: bad arguments");
    CVK_CHECK_ARG(cvk_aligned16(workspace) && workspace_bytes >= cvk_conv3x3_w2d_workspace_bytes(N, H, W, Cin, Cout),
                  "cvk_conv3x3_w2d: workspace too small or misaligned");
    const int T = w2_tiles(4, N, H, W);
    float* const V = (float*)workspace;
    float* const Mo = V + (size_t)36 * w2_tpad(T) * Cin + 128;
    int rc = cvk_w2d_input_transform(x, V, N, H, W, Cin, stream);
    if (rc == CVK_OK) rc = cvk_w2d_gemm(V, U, Mo, T, Cin, Cout, stream);
    if (rc == CVK_OK) rc = cvk_w2d_output(Mo, bias, y, stats, counts, N, H, W, Cin, Cout, ldy, stream);
    return rc;
}

extern "C" int cvk_w2d_wgrad_ksplit(int T, int Cin_pad, int Cout) { return w2i_wgrad_ksplit(4, T, Cin_pad, Cout); }
extern "C" int cvk_w2d_dy_transform(const float* dy, int ld_dy, float* E, int N, int H, int W, int Cout, void* stream) {
    return w2i_dy_transform(4, "cvk_w2d_dy_transform", dy, ld_dy, E, N, H, W, Cout, stream);
}
extern "C" int cvk_w2d_dy_transform_both(const float* dy, int ld_dy, float* Vp, float* E, int N, int H, int W, int Cout, void* stream) {
    return w2i_dy_both(4, "cvk_w2d_dy_transform_both", dy, ld_dy, Vp, E, N, H, W, Cout, stream);
}
extern "C" int cvk_w2d_gemm_tn(const float* E, const float* V, float* P, int T, int Cin_pad, int Cout, void* stream) {
    return w2i_gemm_tn(4, "cvk_w2d_gemm_tn", E, V, P, T, Cin_pad, Cout, stream);
}
extern "C" int cvk_w2d_wgrad_output(const float* P, float* dw, int T, int Cin, int Cin_pad, int Cout, void* stream) {
    return w2i_wgrad_output(4, "cvk_w2d_wgrad_output", P, dw, T, Cin, Cin_pad, Cout, stream);
}
extern "C" size_t cvk_conv3x3_wgrad_w2d_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout) {
    return w2i_wgrad_workspace_bytes(4, N, H, W, Cin_pad, Cout);
}

// x == NULL: the workspace already holds V (cvk_w2d_input_transform of x, e.g. kept from the forward pass) at its start
extern "C" int cvk_conv3x3_wgrad_w2d(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                                     int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(dy && dw && workspace, "cvk_conv3x3_wgrad_w2d: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= Cin_pad && Cin_pad % 4 == 0 && Cout > 0 && Cout % 4 == 0 && ld_dy >= Cout && ld_dy % 4 == 0,
                  "cvk_conv3x3_wgrad_w2d: bad shape (channel counts must be multiples of 4)");
    CVK_CHECK_ARG(cvk_aligned16(workspace) && workspace_bytes >= cvk_conv3x3_wgrad_w2d_workspace_bytes(N, H, W, Cin_pad, Cout),
                  "cvk_conv3x3_wgrad_w2d: workspace too small or misaligned");
    const int T = w2_tiles(4, N, H, W), Tpad = w2_tpad(T);
    float* const V = (float*)workspace;
    float* const E = V + (size_t)36 * Tpad * Cin_pad + 128;
    float* const P = E + (size_t)36 * Tpad * Cout + 128;
    int rc = x ? cvk_w2d_input_transform(x, V, N, H, W, Cin_pad, stream) : CVK_OK;
    if (rc == CVK_OK) rc = cvk_w2d_dy_transform(dy, ld_dy, E, N, H, W, Cout, stream);
    if (rc == CVK_OK) rc = cvk_w2d_gemm_tn(E, V, P, T, Cin_pad, Cout, stream);
    if (rc == CVK_OK) rc = cvk_w2d_wgrad_output(P, dw, T, Cin, Cin_pad, Cout, stream);
    return rc;
}

// ---- C ABI: F(6x6,3x3) — the same pipeline with 64 transform indices on 8 x 8 input tiles ---------------------------------------
extern "C" int cvk_w6_tiles(int N, int H, int W) { return w2i_tiles(6, N, H, W); }
extern "C" int cvk_w6_stat_partials(int N, int H, int W) { return w2i_stat_partials(6, N, H, W); }
extern "C" int cvk_w6_ksplit(int T, int Cin, int Cout) { return w2i_ksplit(6, T, Cin, Cout); }
extern "C" size_t cvk_conv3x3_w6_workspace_bytes(int N, int H, int W, int Cin, int Cout) { return w2i_workspace_bytes(6, N, H, W, Cin, Cout); }
extern "C" int cvk_w6_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream) {
    return w2i_weight_transform(6, "cvk_w6_weight_transform", w, U, Cout, Cin, stream);
}
extern "C" int cvk_w6_weight_transform_dgrad(const float* w, float* U, int Cout, int Cin, void* stream) {
    return w2i_weight_transform_dgrad(6, "cvk_w6_weight_transform_dgrad", w, U, Cout, Cin, stream);
}
extern "C" int cvk_w6_input_transform(const float* x, float* V, int N, int H, int W, int Cin, void* stream) {
    return w2i_input_transform(6, "cvk_w6_input_transform", x, V, N, H, W, Cin, stream);
}
extern "C" int cvk_w6_gemm(const float* V, const float* U, float* Mo, int T, int Cin, int Cout, void* stream) {
    return w2i_gemm(6, "cvk_w6_gemm", V, U, Mo, T, Cin, Cout, stream);
}
extern "C" int cvk_w6_output(const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H, int W, int Cin,
                             int Cout, int ldy, void* stream) {
    return w2i_output(6, "cvk_w6_output", Mo, bias, y, stats, counts, N, H, W, Cin, Cout, ldy, stream);
}
extern "C" int cvk_w6_wgrad_ksplit(int T, int Cin_pad, int Cout) { return w2i_wgrad_ksplit(6, T, Cin_pad, Cout); }
extern "C" int cvk_w6_dy_transform(const float* dy, int ld_dy, float* E, int N, int H, int W, int Cout, void* stream) {
    return w2i_dy_transform(6, "cvk_w6_dy_transform", dy, ld_dy, E, N, H, W, Cout, stream);
}
extern "C" int cvk_w6_dy_transform_both(const float* dy, int ld_dy, float* Vp, float* E, int N, int H, int W, int Cout, void* stream) {
    return w2i_dy_both(6, "cvk_w6_dy_transform_both", dy, ld_dy, Vp, E, N, H, W, Cout, stream);
}
extern "C" int cvk_w6_gemm_tn(const float* E, const float* V, float* P, int T, int Cin_pad, int Cout, void* stream) {
    return w2i_gemm_tn(6, "cvk_w6_gemm_tn", E, V, P, T, Cin_pad, Cout, stream);
}
extern "C" int cvk_w6_wgrad_output(const float* P, float* dw, int T, int Cin, int Cin_pad, int Cout, void* stream) {
    return w2i_wgrad_output(6, "cvk_w6_wgrad_output", P, dw, T, Cin, Cin_pad, Cout, stream);
}
