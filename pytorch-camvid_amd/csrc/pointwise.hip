// pointwise.hip — the HBM-bound data-movement operators of the UNet/SegNet path, NHWC fp32, gfx950:
//   layout import/export (logical NCHW <-> NHWC), F.pad frame zeroing (models/unet.py:120-123),
//   MaxPool2d(2,2) fwd/bwd (models/unet.py:92), MaxPool2d(return_indices)/MaxUnpool2d (models/segnet.py:79-80),
//   bilinear x2 align_corners=True fwd/bwd (models/unet.py:25).
// torch.cat (models/unet.py:124) has no kernel at all: producers write straight into channel slices of the concat
// buffer through cvk_view strides.  Every kernel is one thread per (pixel, 4-channel vector): 16-byte accesses,
// coalesced along the channel dimension; grid-stride loops capped at 16K blocks.
#include "cvk_common.h"

namespace {

inline int grid_for(long total) {
    const long b = (total + 255) / 256;
    return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384);
}

#define CVK_GRID_STRIDE(i, total) \
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------ layout
__global__ void k_import_small(const float* __restrict__ src, int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                               float* __restrict__ dst, int N, int C, int H, int W) {  // C <= 4, ld == 4
    const long total = (long)N * H * W;
    CVK_GRID_STRIDE(i, total) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int y = (int)(t % H), n = (int)(t / H);
        const float* p = src + n * sN + y * sH + x * sW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) v[c] = p[c * sC];
        *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    }
}

__global__ void k_import_generic(const float* __restrict__ src, int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                                 float* __restrict__ dst, int ld, int N, int C, int H, int W) {
    const long total = (long)N * H * W * ld;
    CVK_GRID_STRIDE(i, total) {
        const int c = (int)(i % ld);
        long t = i / ld;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H), n = (int)(t / H);
        dst[i] = c < C ? src[n * sN + c * sC + y * sH + x * sW] : 0.f;
    }
}

__global__ void k_export_generic(const float* __restrict__ src, int ld, float* __restrict__ dst, int64_t dN, int64_t dC,
                                 int64_t dH, int64_t dW, int N, int C, int H, int W) {
    const long total = (long)N * H * W * C;
    CVK_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long t = i / C;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H), n = (int)(t / H);
        dst[n * dN + c * dC + y * dH + x * dW] = src[(((long)n * H + y) * W + x) * ld + c];
    }
}

__global__ void k_zero_frame(cvk_view b, int N, int H, int W, int C, int y0, int x0, int h, int w) {
    const long total = (long)N * H * W * C;
    CVK_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long t = i / C;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H), n = (int)(t / H);
        const bool inside = y >= y0 && y < y0 + h && x >= x0 && x < x0 + w;
        if (!inside) b.ptr[n * b.sN + y * b.sY + x * b.sX + c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ max pool 2x2
template <int V> struct VT_;
template <> struct VT_<4> { typedef f32x4 T; };
template <> struct VT_<1> { typedef float T; };
template <int V> __device__ __forceinline__ float& el(typename VT_<V>::T& v, int j);
template <> __device__ __forceinline__ float& el<4>(f32x4& v, int j) { return reinterpret_cast<float*>(&v)[j]; }
template <> __device__ __forceinline__ float& el<1>(float& v, int) { return v; }

// first maximum in window scan order (0,0),(0,1),(1,0),(1,1); NaN propagates like ATen (v > best || v != v)
template <int V>
__global__ void k_maxpool_fwd(cvk_view x, float* __restrict__ out, uint8_t* __restrict__ code, int N, int H, int W, int C) {
    typedef typename VT_<V>::T VT;
    const int Ho = H / 2, Wo = W / 2, cvn = C / V;
    const long total = (long)N * Ho * Wo * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho), n = (int)(t / Ho);
        const float* p = x.ptr + n * x.sN + (2 * yo) * x.sY + (2 * xo) * x.sX + cv * V;
        VT v[4];
        v[0] = *reinterpret_cast<const VT*>(p);
        v[1] = *reinterpret_cast<const VT*>(p + x.sX);
        v[2] = *reinterpret_cast<const VT*>(p + x.sY);
        v[3] = *reinterpret_cast<const VT*>(p + x.sY + x.sX);
        VT best = v[0];
        uint8_t cd[V];
#pragma unroll
        for (int j = 0; j < V; ++j) cd[j] = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float a = el<V>(v[k], j), b = el<V>(best, j);
                if (a > b || a != a) { el<V>(best, j) = a; cd[j] = (uint8_t)k; }
            }
        const long o = (((long)n * Ho + yo) * Wo + xo) * C + cv * V;
        *reinterpret_cast<VT*>(out + o) = best;
        if (code != nullptr) {
#pragma unroll
            for (int j = 0; j < V; ++j) code[o + j] = cd[j];
        }
    }
}

// cells = ceil(H/2) x ceil(W/2); a cell on the odd trailing row/column has no pooling window (gradient 0).
// SRC: 0 = recompute arg-max from x, 1 = read uint8 code.  Used for pool-backward (value = dout) and for unpool-forward.
template <int V, int SRC>
__global__ void k_pool_scatter(const float* __restrict__ val, cvk_view x, const uint8_t* __restrict__ code, cvk_view dx,
                               int accumulate, int N, int H, int W, int C) {
    typedef typename VT_<V>::T VT;
    const int Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2, cvn = C / V;
    const long total = (long)N * Hc * Wc * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xc = (int)(t % Wc);
        t /= Wc;
        const int yc = (int)(t % Hc), n = (int)(t / Hc);
        float* d = dx.ptr + n * dx.sN + (2 * yc) * dx.sY + (2 * xc) * dx.sX + cv * V;
        const bool full = yc < Ho && xc < Wo;
        if (!full) {
            if (!accumulate) {
                VT z;
#pragma unroll
                for (int j = 0; j < V; ++j) el<V>(z, j) = 0.f;
                for (int dyy = 0; dyy < 2; ++dyy)
                    for (int dxx = 0; dxx < 2; ++dxx)
                        if (2 * yc + dyy < H && 2 * xc + dxx < W) *reinterpret_cast<VT*>(d + dyy * dx.sY + dxx * dx.sX) = z;
            }
            continue;
        }
        const long o = (((long)n * Ho + yc) * Wo + xc) * C + cv * V;
        VT g = *reinterpret_cast<const VT*>(val + o);
        uint8_t cd[V];
        if (SRC == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) cd[j] = code[o + j];
        } else {
            const float* p = x.ptr + n * x.sN + (2 * yc) * x.sY + (2 * xc) * x.sX + cv * V;
            VT v[4];
            v[0] = *reinterpret_cast<const VT*>(p);
            v[1] = *reinterpret_cast<const VT*>(p + x.sX);
            v[2] = *reinterpret_cast<const VT*>(p + x.sY);
            v[3] = *reinterpret_cast<const VT*>(p + x.sY + x.sX);
            VT best = v[0];
#pragma unroll
            for (int j = 0; j < V; ++j) cd[j] = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float a = el<V>(v[k], j), b = el<V>(best, j);
                    if (a > b || a != a) { el<V>(best, j) = a; cd[j] = (uint8_t)k; }
                }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float* q = d + (k >> 1) * dx.sY + (k & 1) * dx.sX;
            VT o4;
            if (accumulate) o4 = *reinterpret_cast<const VT*>(q);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float r = cd[j] == k ? el<V>(g, j) : 0.f;
                el<V>(o4, j) = accumulate ? el<V>(o4, j) + r : r;
            }
            *reinterpret_cast<VT*>(q) = o4;
        }
    }
}

// Round 6: k_pool_scatter<4, SRC> that ALSO leaves the first pass of the producing block's BatchNorm+ReLU backward (csrc/bn.hip k_bn_bwd<MODE 0>:
// sum g and sum g * xhat over the block's output gradient, g = dO where the ReLU passed, xhat = (yP - mean) * rstd).  The max-pool backward is the
// LAST writer of that gradient (the skip connection's share was written earlier), it touches every element of it, and the reduce pass would read
// them all again: here the final values are in registers — one read of yP is added, one read of dO and a launch go.  A block owns a contiguous range of
// cells, a thread a fixed 4-channel vector (256 % (C/4) == 0): partial sums [2][PB][C] for cvk_colsum_finalize, combined in a fixed order.
template <int SRC>
__global__ __launch_bounds__(256) void k_pool_scatter_bnred(const float* __restrict__ val, cvk_view x, const uint8_t* __restrict__ code, cvk_view dx,
                                                           int accumulate, int N, int H, int W, int C, const float* __restrict__ yP, int ldp,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ part, int cpb, int PB) {
    __shared__ float red[2][256 * 4];
    const int Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2, cvn = C / 4;
    const int ppp = 256 / cvn;
    const int t = threadIdx.x, cv = t % cvn, pr = t / cvn, c = cv * 4;
    const long total = (long)N * Hc * Wc;
    const long cbeg = (long)blockIdx.x * cpb, cend = cbeg + cpb < total ? cbeg + cpb : total;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    // two cells per iteration, every load of both issued before the first use: the pass is latency-bound otherwise (one cell per
    // iteration: 4.4 TB/s of its own bytes)
    constexpr int U = 2;
    for (long cell0 = cbeg + pr; cell0 < cend; cell0 += (long)U * ppp) {
        float* d[U];
        bool full[U], live[U];
        int nn[U], yc_[U], xc_[U];
        f32x4 g[U], o4[U][4], yv[U][4], xv[U][4];
        uint8_t cd[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long cell = cell0 + (long)u * ppp;
            live[u] = cell < cend;
            const long cc = live[u] ? cell : cbeg;
            const int xc = (int)(cc % Wc);
            const long tq = cc / Wc;
            const int yc = (int)(tq % Hc), n = (int)(tq / Hc);
            nn[u] = n; yc_[u] = yc; xc_[u] = xc;
            d[u] = dx.ptr + n * dx.sN + (2 * yc) * dx.sY + (2 * xc) * dx.sX + c;
            full[u] = live[u] && yc < Ho && xc < Wo;
            g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) cd[u][j] = 255;
            if (full[u]) {
                const long o = (((long)n * Ho + yc) * Wo + xc) * C + c;
                g[u] = *reinterpret_cast<const f32x4*>(val + o);
                if (SRC == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cd[u][j] = code[o + j];
                } else {
                    const float* p = x.ptr + n * x.sN + (2 * yc) * x.sY + (2 * xc) * x.sX + c;
                    xv[u][0] = *reinterpret_cast<const f32x4*>(p);
                    xv[u][1] = *reinterpret_cast<const f32x4*>(p + x.sX);
                    xv[u][2] = *reinterpret_cast<const f32x4*>(p + x.sY);
                    xv[u][3] = *reinterpret_cast<const f32x4*>(p + x.sY + x.sX);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = 2 * yc + (k >> 1), xx = 2 * xc + (k & 1);
                o4[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                yv[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!live[u] || yy >= H || xx >= W) continue;
                if (accumulate) o4[u][k] = *reinterpret_cast<const f32x4*>(d[u] + (k >> 1) * dx.sY + (k & 1) * dx.sX);
                yv[u][k] = *reinterpret_cast<const f32x4*>(yP + (((size_t)n * H + yy) * W + xx) * ldp + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            if (full[u] && SRC == 0) {
                f32x4 best = xv[u][0];
#pragma unroll
                for (int j = 0; j < 4; ++j) cd[u][j] = 0;
#pragma unroll
                for (int k = 1; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = xv[u][k][j], b = best[j];
                        if (a > b || a != a) { best[j] = a; cd[u][j] = (uint8_t)k; }
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = 2 * yc_[u] + (k >> 1), xx = 2 * xc_[u] + (k & 1);
                if (yy >= H || xx >= W) continue;                 // odd trailing row / column: the cell has fewer than four pixels
                f32x4 o = o4[u][k];
                if (full[u]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += cd[u][j] == k ? g[u][j] : 0.f;
                }
                if (full[u] || !accumulate) *reinterpret_cast<f32x4*>(d[u] + (k >> 1) * dx.sY + (k & 1) * dx.sX) = o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gg = (yv[u][k][j] * sc[j] + sh[j] > 0.f) ? o[j] : 0.f;
                    const float xh = (yv[u][k][j] - mu[j]) * rs[j];
                    s0[j] += gg;
                    s1[j] += gg * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][t * 4 + j] = s0[j]; red[1][t * 4 + j] = s1[j]; }
    __syncthreads();
    if (t < cvn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < ppp; ++p) { a += red[0][(p * cvn + t) * 4 + j]; b += red[1][(p * cvn + t) * 4 + j]; }
            part[(size_t)blockIdx.x * C + t * 4 + j] = a;
            part[(size_t)(PB + blockIdx.x) * C + t * 4 + j] = b;
        }
    }
}

template <int V>
__global__ void k_unpool_bwd(const float* __restrict__ dout, const uint8_t* __restrict__ code, float* __restrict__ dv, int N,
                             int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C;
    CVK_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long t = i / C;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho), n = (int)(t / Ho);
        const int k = code[i];
        dv[i] = dout[(((long)n * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C + c];
    }
}

__global__ void k_code_to_index(const uint8_t* __restrict__ code, int64_t* __restrict__ idx, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * C * Ho * Wo;
    CVK_GRID_STRIDE(i, total) {  // i enumerates NCHW order of the index tensor
        const int xo = (int)(i % Wo);
        long t = i / Wo;
        const int yo = (int)(t % Ho);
        t /= Ho;
        const int c = (int)(t % C), n = (int)(t / C);
        const int k = code[(((long)n * Ho + yo) * Wo + xo) * C + c];
        idx[i] = (int64_t)(2 * yo + (k >> 1)) * W + 2 * xo + (k & 1);
    }
}

// ------------------------------------------------------------------------------------------------ bilinear x2
// ATen align_corners source index: scale = (in-1)/(out-1) in float, src = scale*dst, i0 = (int)src, l1 = src - i0.
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap make_tap(int dst, float scale, int n_in) {
    Tap t;
    const float src = scale * (float)dst;
    t.i0 = (int)src;
    if (t.i0 > n_in - 1) t.i0 = n_in - 1;
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

// Grid: (chunks of one output/input row, row, image).  The flat grid-stride version spent ~3 64-bit divisions per 16-byte
// result on index arithmetic and was VECTOR-ALU bound at 3.2 TB/s (0.40 of the HBM rate); a row per blockIdx.y and one 32-bit
// multiply-high division inside the row leave the taps as the only arithmetic.  Same float operations, same results.
struct RowDiv {         // exact t / d for t * d < 2^32 (conv_tile.h FastDiv)
    unsigned lo, hi_mask;
    explicit RowDiv(unsigned d) {
        const unsigned long long m = 0x100000000ULL / d + 1ULL;
        lo = (unsigned)m;
        hi_mask = (m >> 32) ? 0xFFFFFFFFu : 0u;
    }
    __device__ __forceinline__ unsigned div(unsigned t) const { return __umulhi(t, lo) + (t & hi_mask); }
};

template <int V>
__global__ __launch_bounds__(256) void k_bilinear_fwd(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                     float sy, float sx, RowDiv dcv) {
    typedef typename VT_<V>::T VT;
    const int Wo = 2 * W, cvn = C / V;
    const int yo = blockIdx.y, n = blockIdx.z;
    const unsigned idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (unsigned)(Wo * cvn)) return;
    const int xo = (int)dcv.div(idx), cv = (int)idx - xo * cvn;
    const Tap ty = make_tap(yo, sy, H), tx = make_tap(xo, sx, W);
    const float* b = x + ((long)n * H * W) * C + cv * V;
    VT v00 = *reinterpret_cast<const VT*>(b + ((long)ty.i0 * W + tx.i0) * C);
    VT v01 = *reinterpret_cast<const VT*>(b + ((long)ty.i0 * W + tx.i1) * C);
    VT v10 = *reinterpret_cast<const VT*>(b + ((long)ty.i1 * W + tx.i0) * C);
    VT v11 = *reinterpret_cast<const VT*>(b + ((long)ty.i1 * W + tx.i1) * C);
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j)
        el<V>(o, j) = ty.l0 * (tx.l0 * el<V>(v00, j) + tx.l1 * el<V>(v01, j)) +
                      ty.l1 * (tx.l0 * el<V>(v10, j) + tx.l1 * el<V>(v11, j));
    *reinterpret_cast<VT*>(out + (((long)n * 2 * H + yo) * Wo) * C + (long)idx * V) = o;
}

// LDS-tiled forward for 64-channel chunks: a block = 4 x 32 output pixels x 64 channels.  The flat version reads its four
// taps straight from global memory: 4 x the output bytes through L1/L2 (5.3 GB per step, ~12 TB/s at the texture path) for
// 0.33 GB of input.  Here the <= 4 x 18 input pixels a tile needs are loaded once (16-byte lanes, 256 B per pixel chunk)
// and the taps come from LDS; same float operations, bitwise the same results.
constexpr int BL_TY = 4, BL_TX = 32, BL_IY = 4, BL_IX = 18;
__global__ __launch_bounds__(256) void k_bilinear_fwd_tiled(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                           float sy, float sx, int tilesX) {
    __shared__ f32x4 tile[BL_IY * BL_IX * 16];
    const int Ho = 2 * H, Wo = 2 * W;
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX, c0 = blockIdx.y * 64, n = blockIdx.z;
    const int yo0 = ty * BL_TY, xo0 = tx * BL_TX;
    const int iy0 = make_tap(yo0, sy, H).i0, ix0 = make_tap(xo0, sx, W).i0;       // first input row / column of the tile
    const int t = threadIdx.x, cv = t & 15;
    const float* const xb = x + ((long)n * H * W) * C + c0 + cv * 4;
    for (int p = t >> 4; p < BL_IY * BL_IX; p += 16) {
        const int ry = p / BL_IX, rx = p - ry * BL_IX;
        const int iy = min(iy0 + ry, H - 1), ix = min(ix0 + rx, W - 1);
        tile[p * 16 + cv] = *reinterpret_cast<const f32x4*>(xb + ((long)iy * W + ix) * C);
    }
    __syncthreads();
    for (int p = t >> 4; p < BL_TY * BL_TX; p += 16) {
        const int oy = p / BL_TX, ox = p - oy * BL_TX;
        const int yo = yo0 + oy, xo = xo0 + ox;
        if (yo >= Ho || xo >= Wo) continue;
        const Tap ty_ = make_tap(yo, sy, H), tx_ = make_tap(xo, sx, W);
        const int r0 = (ty_.i0 - iy0) * BL_IX, r1 = (ty_.i1 - iy0) * BL_IX, q0 = tx_.i0 - ix0, q1 = tx_.i1 - ix0;
        const f32x4 v00 = tile[(r0 + q0) * 16 + cv], v01 = tile[(r0 + q1) * 16 + cv];
        const f32x4 v10 = tile[(r1 + q0) * 16 + cv], v11 = tile[(r1 + q1) * 16 + cv];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ty_.l0 * (tx_.l0 * v00[j] + tx_.l1 * v01[j]) + ty_.l1 * (tx_.l0 * v10[j] + tx_.l1 * v11[j]);
        *reinterpret_cast<f32x4*>(out + (((long)n * Ho + yo) * Wo + xo) * C + c0 + cv * 4) = o;
    }
}

// backward = gather over the output pixels whose taps touch input pixel (yi, xi): deterministic, no atomics.  (An LDS-tiled
// version — 4 x 16 input pixels x 32 channels per block, the 12 x 36 gradient pixels loaded once — was 1.7x SLOWER: 1.7x halo
// reads, two blocks per CU, load and compute phases that do not overlap; the re-reads of this version are L2 hits.)
template <int V>
__global__ __launch_bounds__(256) void k_bilinear_bwd(const float* __restrict__ dout, float* __restrict__ dx, int H, int W, int C,
                                                     float sy, float sx, RowDiv dcv, int gx, RowDiv dgx) {
    typedef typename VT_<V>::T VT;
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    // 1-D grid of gx column blocks x (N * H) input rows, dealt to the XCDs in contiguous chunks of rows: neighbouring input rows
    // share two of their ~four gradient rows, and with one row per workgroup in dispatch order those rows were fetched by every
    // XCD's L2 (PMC: 2.5x the gradient tensor crossed the fabric per launch)
    const unsigned lid = (unsigned)cvk_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned row = dgx.div(lid), bx = lid - row * gx;
    const int n = (int)(row / (unsigned)H), yi = (int)(row - (unsigned)n * H);
    const unsigned idx = bx * 256 + threadIdx.x;
    if (idx >= (unsigned)(W * cvn)) return;
    const int xi = (int)dcv.div(idx), cv = (int)idx - xi * cvn;
    // x2 with align_corners: src = s * dst, s = (n-1)/(2n-1) in [1/3, 1/2) — every output row/column whose taps touch input
    // index i lies in [2i-2, 2i+3] (s*(2i-3) < i-1 and s*(2i+4) >= i+1 for all n >= 2; n == 1: s = 0, both outputs map to 0).
    // A fixed 6 x 6 candidate window with per-candidate weights (zero = not a tap of this pixel) replaces the data-dependent
    // loop bounds of the first version: the loads of all contributing candidates are in flight together (the old loop had one).
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int yo = 2 * yi - 2 + k, xo = 2 * xi - 2 + k;
        wy[k] = 0.f; wx[k] = 0.f;
        if ((unsigned)yo < (unsigned)Ho) {
            const Tap t = make_tap(yo, sy, H);
            wy[k] = (t.i0 == yi ? t.l0 : 0.f) + (t.i1 == yi ? t.l1 : 0.f);
        }
        if ((unsigned)xo < (unsigned)Wo) {
            const Tap t = make_tap(xo, sx, W);
            wx[k] = (t.i0 == xi ? t.l0 : 0.f) + (t.i1 == xi ? t.l1 : 0.f);
        }
    }
    VT acc;
#pragma unroll
    for (int j = 0; j < V; ++j) el<V>(acc, j) = 0.f;
    const float* b = dout + ((long)n * Ho * Wo) * C + cv * V;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {                             // block-uniform
        if (wy[ky] == 0.f) continue;
        const int yo = 2 * yi - 2 + ky;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            if (wx[kx] == 0.f) continue;
            const int xo = 2 * xi - 2 + kx;
            VT g = *reinterpret_cast<const VT*>(b + ((long)yo * Wo + xo) * C);
            const float w = wy[ky] * wx[kx];
#pragma unroll
            for (int j = 0; j < V; ++j) el<V>(acc, j) += w * el<V>(g, j);
        }
    }
    *reinterpret_cast<VT*>(dx + (((long)n * H + yi) * W) * C + (long)idx * V) = acc;
}


// Round 6: the same gather, SEPARABLE and walked down the image.  x2 bilinear upsampling is  out = Ry x Rx^T  with two taps per output row / column, so
// dx = Ry^T (dout Rx).  A thread owns one input column xi (4 channels) of a strip of input rows and walks the output rows that feed the strip: per output
// row it forms h = sum_kx wx[kx] dout[yo][2 xi - 2 + kx] (the <= 4 columns with a non-zero weight) and adds l0 h to input row i0(yo), l1 h to row
// i0(yo) + 1 — two running accumulators, a row is written when the walk leaves it.  Loads per input pixel: 2 output rows x <= 4 instead of <= 4 x 4
// (every output pixel is read by two threads instead of four: half the L1 / L2 traffic of k_bilinear_bwd, which ran at 3.9 TB/s of HBM traffic with
// no redundant HBM bytes).  Strips overlap by the five output rows of their borders.  Same taps (make_tap = ATen's source index), another summation order.
template <int V>
__global__ __launch_bounds__(256) void k_bilinear_bwd_walk(const float* __restrict__ dout, float* __restrict__ dx, int H, int W, int C,
                                                          float sy, float sx, RowDiv dcv, int gx, RowDiv dgx, int RS, int nstrips) {
    typedef typename VT_<V>::T VT;
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const unsigned lid = (unsigned)cvk_xcd_remap(blockIdx.x, gridDim.x);      // (image, strip) major, column chunk minor: neighbouring strips share an L2
    const unsigned srow = dgx.div(lid), bx = lid - srow * gx;
    const int n = (int)(srow / (unsigned)nstrips), st = (int)(srow - (unsigned)n * nstrips);
    const unsigned idx = bx * 256 + threadIdx.x;
    if (idx >= (unsigned)(W * cvn)) return;
    const int xi = (int)dcv.div(idx), cv = (int)idx - xi * cvn;
    const int r0 = st * RS, r1 = min(H, r0 + RS);
    float wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int xo = 2 * xi - 2 + k;
        wx[k] = 0.f;
        if ((unsigned)xo < (unsigned)Wo) {
            const Tap t = make_tap(xo, sx, W);
            wx[k] = (t.i0 == xi ? t.l0 : 0.f) + (t.i1 == xi ? t.l1 : 0.f);
        }
    }
    const float* b = dout + ((long)n * Ho * Wo) * C + cv * V;
    float* o = dx + ((long)n * H * W) * C + (long)idx * V;
    const int y_beg = max(0, 2 * r0 - 2), y_end = min(Ho - 1, 2 * (r1 - 1) + 3);
    VT a0, a1;
#pragma unroll
    for (int j = 0; j < V; ++j) { el<V>(a0, j) = 0.f; el<V>(a1, j) = 0.f; }
    int cur = make_tap(y_beg, sy, H).i0;                  // input row of a0 (a1: cur + 1)
    auto hrow = [&](int yo) {
        VT h;
#pragma unroll
        for (int j = 0; j < V; ++j) el<V>(h, j) = 0.f;
        const float* p = b + ((long)yo * Wo + (2 * xi - 2)) * C;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (wx[k] == 0.f) continue;
            VT g = *reinterpret_cast<const VT*>(p + (long)k * C);
#pragma unroll
            for (int j = 0; j < V; ++j) el<V>(h, j) += wx[k] * el<V>(g, j);
        }
        return h;
    };
    // two output rows per iteration: their loads are issued together
    for (int yo = y_beg; yo <= y_end; yo += 2) {
        const bool two = yo + 1 <= y_end;
        VT h0 = hrow(yo);
        VT h1 = h0;
        if (two) h1 = hrow(yo + 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const Tap t = make_tap(yo + u, sy, H);          // uniform
            VT h = u == 0 ? h0 : h1;
            while (cur < t.i0) {                            // the walk leaves row cur: it is complete
                if (cur >= r0 && cur < r1) *reinterpret_cast<VT*>(o + (long)cur * W * C) = a0;
                a0 = a1;
#pragma unroll
                for (int j = 0; j < V; ++j) el<V>(a1, j) = 0.f;
                ++cur;
            }
            const float w1 = t.i1 > t.i0 ? t.l1 : 0.f, w0 = t.i1 > t.i0 ? t.l0 : t.l0 + t.l1;      // clamped at the last row: both taps are row i0
#pragma unroll
            for (int j = 0; j < V; ++j) { el<V>(a0, j) += w0 * el<V>(h, j); el<V>(a1, j) += w1 * el<V>(h, j); }
        }
    }
    if (cur >= r0 && cur < r1) *reinterpret_cast<VT*>(o + (long)cur * W * C) = a0;
    if (cur + 1 >= r0 && cur + 1 < r1) *reinterpret_cast<VT*>(o + (long)(cur + 1) * W * C) = a1;
}

inline bool v4ok(int C, const void* a, const void* b) { return C % 4 == 0 && cvk_aligned16(a) && cvk_aligned16(b); }
inline bool view4(const cvk_view& v) { return cvk_aligned16(v.ptr) && ((v.sN | v.sY | v.sX) & 3) == 0; }

}  // namespace

extern "C" int cvk_import_nchw(const float* src, int64_t sN, int64_t sC, int64_t sH, int64_t sW, float* dst, int ld, int N,
                               int C, int H, int W, void* stream) {
    CVK_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "cvk_import_nchw: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (C <= 4 && ld == 4 && cvk_aligned16(dst))
        hipLaunchKernelGGL(k_import_small, dim3(grid_for((long)N * H * W)), dim3(256), 0, s, src, sN, sC, sH, sW, dst, N, C, H, W);
    else
        hipLaunchKernelGGL(k_import_generic, dim3(grid_for((long)N * H * W * ld)), dim3(256), 0, s, src, sN, sC, sH, sW, dst, ld, N, C, H, W);
    CVK_LAUNCH_RETURN("cvk_import_nchw");
}

extern "C" int cvk_export_nchw(const float* src, int ld, float* dst, int64_t dN, int64_t dC, int64_t dH, int64_t dW, int N, int C,
                               int H, int W, void* stream) {
    CVK_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "cvk_export_nchw: bad arguments");
    hipLaunchKernelGGL(k_export_generic, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, src, ld, dst, dN, dC, dH, dW, N, C, H, W);
    CVK_LAUNCH_RETURN("cvk_export_nchw");
}

extern "C" int cvk_zero_frame(cvk_view buf, int N, int H, int W, int C, int y0, int x0, int h, int w, void* stream) {
    CVK_CHECK_ARG(buf.ptr && N > 0 && H > 0 && W > 0 && C > 0 && y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W, "cvk_zero_frame: bad arguments");
    hipLaunchKernelGGL(k_zero_frame, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, buf, N, H, W, C, y0, x0, h, w);
    CVK_LAUNCH_RETURN("cvk_zero_frame");
}

extern "C" int cvk_maxpool2x2_fwd(cvk_view x, float* out, uint8_t* code, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(x.ptr && out && N > 0 && H >= 2 && W >= 2 && C > 0, "cvk_maxpool2x2_fwd: bad arguments (need H,W >= 2)");
    const long cells = (long)N * (H / 2) * (W / 2);
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0 && view4(x) && cvk_aligned16(out))
        hipLaunchKernelGGL(k_maxpool_fwd<4>, dim3(grid_for(cells * (C / 4))), dim3(256), 0, s, x, out, code, N, H, W, C);
    else
        hipLaunchKernelGGL(k_maxpool_fwd<1>, dim3(grid_for(cells * C)), dim3(256), 0, s, x, out, code, N, H, W, C);
    CVK_LAUNCH_RETURN("cvk_maxpool2x2_fwd");
}

extern "C" int cvk_maxpool2x2_bwd(const float* dout, cvk_view x, const uint8_t* code, cvk_view dx, int accumulate, int N, int H,
                                  int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && dx.ptr && (code || x.ptr) && N > 0 && H >= 2 && W >= 2 && C > 0, "cvk_maxpool2x2_bwd: bad arguments");
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = C % 4 == 0 && view4(dx) && cvk_aligned16(dout) && (code || view4(x));
    if (code) {
        if (v4) hipLaunchKernelGGL((k_pool_scatter<4, 1>), dim3(grid_for(cells * (C / 4))), dim3(256), 0, s, dout, x, code, dx, accumulate, N, H, W, C);
        else hipLaunchKernelGGL((k_pool_scatter<1, 1>), dim3(grid_for(cells * C)), dim3(256), 0, s, dout, x, code, dx, accumulate, N, H, W, C);
    } else {
        if (v4) hipLaunchKernelGGL((k_pool_scatter<4, 0>), dim3(grid_for(cells * (C / 4))), dim3(256), 0, s, dout, x, code, dx, accumulate, N, H, W, C);
        else hipLaunchKernelGGL((k_pool_scatter<1, 0>), dim3(grid_for(cells * C)), dim3(256), 0, s, dout, x, code, dx, accumulate, N, H, W, C);
    }
    CVK_LAUNCH_RETURN("cvk_maxpool2x2_bwd");
}

// blocks (= partial sums per channel) of cvk_maxpool2x2_bwd_bnred; 0: the shape is not supported (C/4 must divide 256)
extern "C" int cvk_maxpool2x2_bwd_bnred_blocks(int N, int H, int W, int C) {
    if (N <= 0 || H < 2 || W < 2 || C < 4 || C % 4 || C > 1024 || 256 % (C / 4)) return 0;
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const int ppp = 256 / (C / 4);
#ifndef CVK_PBN_CPT
#define CVK_PBN_CPT 4
#endif
#ifndef CVK_PBN_CAP
#define CVK_PBN_CAP 4096
#endif
    const long want = (cells + (long)CVK_PBN_CPT * ppp - 1) / ((long)CVK_PBN_CPT * ppp);          // >= CVK_PBN_CPT cells per thread
    return (int)(want < 1 ? 1 : (want > CVK_PBN_CAP ? CVK_PBN_CAP : want));
}

// cvk_maxpool2x2_bwd (4-channel vector layout required) that also leaves the partial sums of the producing block's BatchNorm-backward reduce pass
// over the FINISHED gradient dx: part = float[2][cvk_maxpool2x2_bwd_bnred_blocks][C] for cvk_colsum_finalize(part, blocks, C, dbeta, dgamma).
// yP [N*H*W][ldp] = the block's conv output, scale / shift / mean / rstd its BatchNorm constants (contract of cvk_bn_bwd_reduce).
extern "C" int cvk_maxpool2x2_bwd_bnred(const float* dout, cvk_view x, const uint8_t* code, cvk_view dx, int accumulate, int N, int H, int W, int C,
                                        const float* yP, int ldp, const float* scale, const float* shift, const float* mean, const float* rstd,
                                        float* part, void* stream) {
    CVK_CHECK_ARG(dout && dx.ptr && (code || x.ptr) && yP && scale && shift && mean && rstd && part, "cvk_maxpool2x2_bwd_bnred: null pointer");
    const int PB = cvk_maxpool2x2_bwd_bnred_blocks(N, H, W, C);
    CVK_CHECK_ARG(PB > 0 && ldp >= C && ldp % 4 == 0, "cvk_maxpool2x2_bwd_bnred: unsupported shape (C=%d: C/4 must divide 256)", C);
    CVK_CHECK_ARG(view4(dx) && cvk_aligned16(dout) && (code || view4(x)) && cvk_aligned16(yP) && cvk_aligned16(scale) && cvk_aligned16(shift) &&
                  cvk_aligned16(mean) && cvk_aligned16(rstd), "cvk_maxpool2x2_bwd_bnred: needs the 4-channel vector layout");
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const int cpb = (int)((cells + PB - 1) / PB);
    hipStream_t s = (hipStream_t)stream;
    if (code) hipLaunchKernelGGL(k_pool_scatter_bnred<1>, dim3(PB), dim3(256), 0, s, dout, x, code, dx, accumulate, N, H, W, C, yP, ldp, scale, shift, mean, rstd, part, cpb, PB);
    else hipLaunchKernelGGL(k_pool_scatter_bnred<0>, dim3(PB), dim3(256), 0, s, dout, x, code, dx, accumulate, N, H, W, C, yP, ldp, scale, shift, mean, rstd, part, cpb, PB);
    CVK_LAUNCH_RETURN("cvk_maxpool2x2_bwd_bnred");
}

extern "C" int cvk_maxunpool2x2_fwd(const float* v, const uint8_t* code, float* out, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(v && code && out && N > 0 && H >= 2 && W >= 2 && C > 0, "cvk_maxunpool2x2_fwd: bad arguments");
    cvk_view o;
    o.ptr = out; o.sX = C; o.sY = (int64_t)W * C; o.sN = (int64_t)H * W * C;
    cvk_view none;
    none.ptr = nullptr; none.sN = none.sY = none.sX = 0;
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0 && cvk_aligned16(v) && cvk_aligned16(out))
        hipLaunchKernelGGL((k_pool_scatter<4, 1>), dim3(grid_for(cells * (C / 4))), dim3(256), 0, s, v, none, code, o, 0, N, H, W, C);
    else
        hipLaunchKernelGGL((k_pool_scatter<1, 1>), dim3(grid_for(cells * C)), dim3(256), 0, s, v, none, code, o, 0, N, H, W, C);
    CVK_LAUNCH_RETURN("cvk_maxunpool2x2_fwd");
}

extern "C" int cvk_maxunpool2x2_bwd(const float* dout, const uint8_t* code, float* dv, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && code && dv && N > 0 && H >= 2 && W >= 2 && C > 0, "cvk_maxunpool2x2_bwd: bad arguments");
    hipLaunchKernelGGL(k_unpool_bwd<1>, dim3(grid_for((long)N * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream, dout, code, dv, N, H, W, C);
    CVK_LAUNCH_RETURN("cvk_maxunpool2x2_bwd");
}

extern "C" int cvk_pool_code_to_index(const uint8_t* code, int64_t* idx, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(code && idx && N > 0 && H >= 2 && W >= 2 && C > 0, "cvk_pool_code_to_index: bad arguments");
    hipLaunchKernelGGL(k_code_to_index, dim3(grid_for((long)N * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream, code, idx, N, H, W, C);
    CVK_LAUNCH_RETURN("cvk_pool_code_to_index");
}

static inline float ac_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

extern "C" int cvk_bilinear_up2_fwd(const float* x, float* out, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0, "cvk_bilinear_up2_fwd: bad arguments");
    const float sy = ac_scale(H, 2 * H), sx = ac_scale(W, 2 * W);
    hipStream_t s = (hipStream_t)stream;
    CVK_CHECK_ARG(2 * H <= 65535 && N <= 65535 && (long)2 * W * C < (1L << 31), "cvk_bilinear_up2_fwd: frame too large for the row grid");
    if (v4ok(C, x, out) && C % 64 == 0 && H >= 2 && W >= 2 && (long)cvk_cdiv(2 * H, BL_TY) * cvk_cdiv(2 * W, BL_TX) < (1L << 31) && C / 64 <= 65535) {
        const int tilesX = cvk_cdiv(2 * W, BL_TX), tilesY = cvk_cdiv(2 * H, BL_TY);
        hipLaunchKernelGGL(k_bilinear_fwd_tiled, dim3(tilesX * tilesY, C / 64, N), dim3(256), 0, s, x, out, H, W, C, sy, sx, tilesX);
    } else if (v4ok(C, x, out))
        hipLaunchKernelGGL(k_bilinear_fwd<4>, dim3(cvk_cdiv((long)2 * W * (C / 4), 256), 2 * H, N), dim3(256), 0, s, x, out, H, W, C, sy, sx, RowDiv(C / 4));
    else
        hipLaunchKernelGGL(k_bilinear_fwd<1>, dim3(cvk_cdiv((long)2 * W * C, 256), 2 * H, N), dim3(256), 0, s, x, out, H, W, C, sy, sx, RowDiv(C));
    CVK_LAUNCH_RETURN("cvk_bilinear_up2_fwd");
}

extern "C" int cvk_bilinear_up2_bwd(const float* dout, float* dx, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && dx && N > 0 && H > 0 && W > 0 && C > 0, "cvk_bilinear_up2_bwd: bad arguments");
    const float sy = ac_scale(H, 2 * H), sx = ac_scale(W, 2 * W);
    hipStream_t s = (hipStream_t)stream;
    CVK_CHECK_ARG((long)2 * W * C < (1L << 31), "cvk_bilinear_up2_bwd: frame too large for the row grid");
    const bool v4 = v4ok(C, dout, dx);
    const int gx = cvk_cdiv((long)W * (v4 ? C / 4 : C), 256);
    CVK_CHECK_ARG((long)gx * H * N < (1L << 31) - 8, "cvk_bilinear_up2_bwd: too many workgroups");
    if (v4 && H >= 2) {
        // strips: enough threads to fill the chip, at least 4 input rows each (the five halo rows are re-read per strip).  Four launches of the headline
        // step, interleaved builds on one box: one input row per workgroup row (k_bilinear_bwd) 0.433 ms; the walk with >= 200k / 400k / 800k threads
        // 0.448 / 0.390 / 0.419 ms.  (The same walk for the bf16 tensors of configs[3] measured SLOWER, 0.545 against 0.49 ms: not taken there.)
        const long cols = (long)N * W * (C / 4);
        long want = (400000 + cols - 1) / cols;
        int RS = (int)(H / (want < 1 ? 1 : want));
        RS = RS < 4 ? 4 : RS;
        RS = RS > H ? H : RS;
        const int nstrips = cvk_cdiv(H, RS);
        const dim3 gridw((unsigned)((long)gx * nstrips * N));
        hipLaunchKernelGGL(k_bilinear_bwd_walk<4>, gridw, dim3(256), 0, s, dout, dx, H, W, C, sy, sx, RowDiv(C / 4), gx, RowDiv(gx), RS, nstrips);
        CVK_LAUNCH_RETURN("cvk_bilinear_up2_bwd");
    }
    const dim3 grid((unsigned)((long)gx * H * N));
    if (v4)
        hipLaunchKernelGGL(k_bilinear_bwd<4>, grid, dim3(256), 0, s, dout, dx, H, W, C, sy, sx, RowDiv(C / 4), gx, RowDiv(gx));
    else
        hipLaunchKernelGGL(k_bilinear_bwd<1>, grid, dim3(256), 0, s, dout, dx, H, W, C, sy, sx, RowDiv(C), gx, RowDiv(gx));
    CVK_LAUNCH_RETURN("cvk_bilinear_up2_bwd");
}
