// elem_bf16.hip — the HBM-bound operators of the bf16-storage path (BASELINE.json configs[3]): every activation and
// activation-gradient tensor lives in HBM as bf16 NHWC, so each pass moves half the bytes of its fp32 twin
// (bn.hip / pointwise.hip); arithmetic is fp32 in registers.  Reference sites as there:
//   BatchNorm2d + ReLU forward/backward   models/unet.py:12-13        MaxPool2d(2,2)   models/unet.py:92
//   bilinear x2 (align_corners=True)      models/unet.py:25           NCHW import      train.py:126-128
// One thread owns a vector of V channels of one pixel: V = 8 (16-byte bf16 accesses) or 4 (the 12-class head).
// The 2x2 max pool of the encoder is FUSED into the BN-apply pass (the pass that produces the skip tensor also writes
// the pooled tensor: the pool never re-reads its input).  The network's logits leave this path as fp32 (OUT_F32) and
// the loss gradient enters it as fp32 (DOUT_F32), so the cross-entropy kernels are shared with the fp32 path.
#include "cvk_common.h"

namespace {

inline int grid_for(long total) {
    const long b = (total + 255) / 256;
    return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384);
}
#define CVK_GRID_STRIDE(i, total) \
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int V> struct FV { float v[V]; };

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned bf_pack(float a, float b) {
    const bf16x2 p = {(__bf16)a, (__bf16)b};      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned, p);
}

template <int V> __device__ __forceinline__ FV<V> load_bf16(const __bf16* p);
template <> __device__ __forceinline__ FV<8> load_bf16<8>(const __bf16* p) {
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
    FV<8> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = bf_lo(w[i]); r.v[2 * i + 1] = bf_hi(w[i]); }
    return r;
}
template <> __device__ __forceinline__ FV<4> load_bf16<4>(const __bf16* p) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(p);
    FV<4> r;
#pragma unroll
    for (int i = 0; i < 2; ++i) { r.v[2 * i] = bf_lo(w[i]); r.v[2 * i + 1] = bf_hi(w[i]); }
    return r;
}
template <int V> __device__ __forceinline__ FV<V> load_bf16_nt(const __bf16* p) { return load_bf16<V>(p); }
// Streaming hints (round 4).  A tensor that is read once more and then dead (the pre-BN y in the apply and dx passes, the incoming gradient in
// the dx pass; in the reduce pass when the pair does not fit the 256 MB Infinity Cache anyway: >= 128 MB each) is loaded non-temporally, so
// that it does not push out what the NEXT kernel reads (the activation just written, dy for the data-grad and weight-grad).  configs[3]:
// +1.2 % images/s (189.9 -> 192.1, 192.2 -> 194.6 on two boxes, interleaved runs; CVK_STREAM_HINTS=0 switches them off).
template <> __device__ __forceinline__ FV<8> load_bf16_nt<8>(const __bf16* p) {
    const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    FV<8> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = bf_lo(w[i]); r.v[2 * i + 1] = bf_hi(w[i]); }
    return r;
}

template <int V> __device__ __forceinline__ void store_bf16(__bf16* p, const FV<V>& f);
template <> __device__ __forceinline__ void store_bf16<8>(__bf16* p, const FV<8>& f) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = bf_pack(f.v[2 * i], f.v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = w;
}
template <> __device__ __forceinline__ void store_bf16<4>(__bf16* p, const FV<4>& f) {
    u32x2 w;
#pragma unroll
    for (int i = 0; i < 2; ++i) w[i] = bf_pack(f.v[2 * i], f.v[2 * i + 1]);
    *reinterpret_cast<u32x2*>(p) = w;
}
template <int V> __device__ __forceinline__ FV<V> load_f32(const float* p) {
    FV<V> r;
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[4 * i + j] = w[j];
    }
    return r;
}
template <int V> __device__ __forceinline__ void store_f32(float* p, const FV<V>& f) {
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
        const f32x4 w = {f.v[4 * i], f.v[4 * i + 1], f.v[4 * i + 2], f.v[4 * i + 3]};
        *reinterpret_cast<f32x4*>(p + 4 * i) = w;
    }
}
// element-typed access through a void pointer: F32 selects the fp32 boundary tensors (logits, loss gradient)
template <int V, bool F32> __device__ __forceinline__ FV<V> load_any(const void* base, int64_t off) {
    if (F32) return load_f32<V>(reinterpret_cast<const float*>(base) + off);
    return load_bf16<V>(reinterpret_cast<const __bf16*>(base) + off);
}
template <int V, bool F32> __device__ __forceinline__ void store_any(void* base, int64_t off, const FV<V>& f) {
    if (F32) store_f32<V>(reinterpret_cast<float*>(base) + off, f);
    else store_bf16<V>(reinterpret_cast<__bf16*>(base) + off, f);
}

struct PixMapH {  // pixel m (row-major over N,H,W) -> element offset inside a strided view
    int64_t sN, sY, sX;
    int H, W, linear;
    __device__ __forceinline__ int64_t off(int m) const {
        if (linear) return (int64_t)m * sX;
        const int hw = H * W;
        const int n = m / hw, rem = m - n * hw;
        const int y = rem / W, x = rem - y * W;
        return n * sN + y * sY + x * sX;
    }
};
PixMapH make_map(const cvk_viewh& v, int H, int W) {
    PixMapH p;
    p.sN = v.sN; p.sY = v.sY; p.sX = v.sX; p.H = H; p.W = W;
    p.linear = (v.sY == (int64_t)W * v.sX && v.sN == (int64_t)H * v.sY) ? 1 : 0;
    return p;
}

// ------------------------------------------------------------------------------------------------ NCHW fp32 -> NHWC bf16
__global__ void k_import_bf16(const float* __restrict__ src, int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                              __bf16* __restrict__ dst, int ld, int N, int C, int H, int W) {
    const int gv = ld / 8;
    const long total = (long)N * H * W * gv;
    CVK_GRID_STRIDE(i, total) {
        const int g = (int)(i % gv);
        long t = i / gv;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H), n = (int)(t / H);
        const float* p = src + n * sN + y * sH + x * sW;
        FV<8> f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g * 8 + j;
            f.v[j] = c < C ? p[c * sC] : 0.f;
        }
        store_bf16<8>(dst + i * 8, f);
    }
}

// ------------------------------------------------------------------------------------------------ BN apply + ReLU (+ 2x2 max pool)
// out = relu(y * scale + shift) through a strided view; POOL: one thread owns a 2x2 cell and also writes its maximum
// (ties, NaN: the maximum of the four VALUES, which is all the forward needs; the backward recomputes the arg-max with
// the first-maximum rule of ATen).  Cells cover ceil(H/2) x ceil(W/2); cells on an odd trailing row/column write no pool.
template <int V, bool OUT_F32, bool POOL, bool NT = false>
__global__ __launch_bounds__(256) void k_apply_bf16(const __bf16* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, void* __restrict__ out, PixMapH om,
                                                   __bf16* __restrict__ pool, int N, int H, int W, int C) {
    const int cvn = C / V;
    if (!POOL) {
        const long total = (long)N * H * W * cvn;
        CVK_GRID_STRIDE(i, total) {
            const int m = (int)(i / cvn);
            const int c = (int)(i - (long)m * cvn) * V;
            const FV<V> v = NT ? load_bf16_nt<V>(y + (size_t)m * ldy + c) : load_bf16<V>(y + (size_t)m * ldy + c);
            const FV<V> sc = load_f32<V>(scale + c), sh = load_f32<V>(shift + c);
            FV<V> o;
#pragma unroll
            for (int j = 0; j < V; ++j) o.v[j] = fmaxf(v.v[j] * sc.v[j] + sh.v[j], 0.f);
            store_any<V, OUT_F32>(out, om.off(m) + c, o);
        }
    } else {
        const int Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
        const long total = (long)N * Hc * Wc * cvn;
        CVK_GRID_STRIDE(i, total) {
            const int cv = (int)(i % cvn);
            long t = i / cvn;
            const int xc = (int)(t % Wc);
            t /= Wc;
            const int yc = (int)(t % Hc), n = (int)(t / Hc);
            const int c = cv * V;
            const FV<V> sc = load_f32<V>(scale + c), sh = load_f32<V>(shift + c);
            FV<V> best;
#pragma unroll
            for (int j = 0; j < V; ++j) best.v[j] = 0.f;              // post-ReLU values are >= 0
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = 2 * yc + (k >> 1), xx = 2 * xc + (k & 1);
                if (yy < H && xx < W) {
                    const int m = (n * H + yy) * W + xx;
                    const FV<V> v = NT ? load_bf16_nt<V>(y + (size_t)m * ldy + c) : load_bf16<V>(y + (size_t)m * ldy + c);
                    FV<V> o;
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        o.v[j] = fmaxf(v.v[j] * sc.v[j] + sh.v[j], 0.f);
                        best.v[j] = (o.v[j] > best.v[j] || o.v[j] != o.v[j]) ? o.v[j] : best.v[j];
                    }
                    store_any<V, OUT_F32>(out, om.off(m) + c, o);
                }
            }
            if (yc < Ho && xc < Wo) {
                // the pooled tensor holds the maximum of the ROUNDED (stored) activations
#pragma unroll
                for (int j = 0; j < V; ++j) best.v[j] = (float)(__bf16)best.v[j];
                store_bf16<V>(pool + (((size_t)n * Ho + yc) * Wo + xc) * C + c, best);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ BN + ReLU backward
// Walker as in bn.hip: block b owns pixel rows [b*rows, (b+1)*rows); thread t owns channel vector (t % cvn) and walks
// pixels.  MODE 0: partial sums of g and g*xhat (g = dout masked by the ReLU).  MODE 1: dy = scale*(g - dbeta/M -
// xhat*dgamma/M) written as bf16 rows of pitch ld_dy (columns C..ld_dy-1 zero), partial column sums of dy (conv bias grad).
template <int V, int MODE, bool DOUT_F32, bool NT = false>
__global__ __launch_bounds__(256) void k_bnbwd_bf16(const void* __restrict__ dout, PixMapH dm, const __bf16* __restrict__ y, int ldy,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                   __bf16* __restrict__ dy, int ld_dy, float* __restrict__ part, int M, int C,
                                                   int rows, int PB, int use_batch_stats) {
    __shared__ float red[2][256 * V];
    const int cvn = C / V;
    const int ppp = 256 / cvn;
    const int t = threadIdx.x;
    const bool active = t < cvn * ppp;
    const int cv = t % cvn, pr = t / cvn;
    const int c = cv * V;
    const int mbeg = blockIdx.x * rows, mend = min(M, mbeg + rows);
    float s0[V], s1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (active) {
        const FV<V> sc = load_f32<V>(scale + c), sh = load_f32<V>(shift + c), mu = load_f32<V>(mean + c), rs = load_f32<V>(rstd + c);
        float k1[V], k2[V];
        if (MODE == 1) {
            const float invM = 1.f / (float)M;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                k1[j] = use_batch_stats ? dbeta[c + j] * invM : 0.f;
                k2[j] = use_batch_stats ? dgamma[c + j] * invM : 0.f;
            }
        }
#pragma unroll 4
        for (int m = mbeg + pr; m < mend; m += ppp) {     // four iterations' loads in flight (the sums stay in order): 32 B per
            // thread and iteration, 4 blocks per CU — one iteration in flight is half the ~64 KiB per CU that 8 TB/s x 2 us asks for
            const FV<V> d = (NT && !DOUT_F32) ? load_bf16_nt<V>(reinterpret_cast<const __bf16*>(dout) + dm.off(m) + c) : load_any<V, DOUT_F32>(dout, dm.off(m) + c);
            const FV<V> yy = NT ? load_bf16_nt<V>(y + (size_t)m * ldy + c) : load_bf16<V>(y + (size_t)m * ldy + c);
            FV<V> o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float g = (yy.v[j] * sc.v[j] + sh.v[j] > 0.f) ? d.v[j] : 0.f;
                const float xh = (yy.v[j] - mu.v[j]) * rs.v[j];
                if (MODE == 0) {
                    s0[j] += g;
                    s1[j] += g * xh;
                } else {
                    const float r = sc.v[j] * (g - k1[j] - xh * k2[j]);
                    o.v[j] = r;
                    s0[j] += r;
                }
            }
            if (MODE == 1) store_bf16<V>(dy + (size_t)m * ld_dy + c, o);
        }
    }
    if (MODE == 1 && ld_dy > C) {                 // zero the channel padding of dy (the data-grad GEMM reads it)
        const int padv = (ld_dy - C) / 4;          // 8-byte groups
        for (int i = t; i < (mend - mbeg) * padv; i += 256) {
            const int m = mbeg + i / padv, g = i % padv;
            *reinterpret_cast<u32x2*>(dy + (size_t)m * ld_dy + C + 4 * g) = u32x2{0u, 0u};
        }
    }
    if (part == nullptr) return;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][t * V + j] = s0[j];
        if (MODE == 0) red[1][t * V + j] = s1[j];
    }
    __syncthreads();
    if (t < cvn) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < ppp; ++p) {
                a += red[0][(p * cvn + t) * V + j];
                if (MODE == 0) b += red[1][(p * cvn + t) * V + j];
            }
            part[(size_t)blockIdx.x * C + t * V + j] = a;
            if (MODE == 0) part[(size_t)(PB + blockIdx.x) * C + t * V + j] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------ max pool backward
// cells = ceil(H/2) x ceil(W/2); arg-max recomputed from the stored activations (first maximum in scan order (0,0),(0,1),
// (1,0),(1,1), NaN wins: ATen's rule); accumulate != 0 adds to dx (the skip half of the concat-buffer gradient).
template <int V>
__global__ void k_pool_bwd_bf16(const __bf16* __restrict__ dout, const __bf16* __restrict__ x, PixMapH xm, __bf16* __restrict__ dx,
                                PixMapH dm, int accumulate, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2, cvn = C / V;
    const long total = (long)N * Hc * Wc * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xc = (int)(t % Wc);
        t /= Wc;
        const int yc = (int)(t % Hc), n = (int)(t / Hc);
        const int c = cv * V;
        const bool full = yc < Ho && xc < Wo;
        FV<V> g;
        int cd[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { g.v[j] = 0.f; cd[j] = -1; }
        if (full) {
            g = load_bf16<V>(dout + (((size_t)n * Ho + yc) * Wo + xc) * C + c);
            FV<V> best;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = (n * H + 2 * yc + (k >> 1)) * W + 2 * xc + (k & 1);
                const FV<V> v = load_bf16<V>(x + xm.off(m) + c);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (k == 0 || v.v[j] > best.v[j] || v.v[j] != v.v[j]) { best.v[j] = v.v[j]; cd[j] = k; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = 2 * yc + (k >> 1), xx = 2 * xc + (k & 1);
            if (yy >= H || xx >= W) continue;
            const int m = (n * H + yy) * W + xx;
            __bf16* q = dx + dm.off(m) + c;
            FV<V> o;
            if (accumulate) o = load_bf16<V>(q);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float r = cd[j] == k ? g.v[j] : 0.f;
                o.v[j] = accumulate ? o.v[j] + r : r;
            }
            store_bf16<V>(q, o);
        }
    }
}


// Round 6 (bf16 twin of csrc/pointwise.hip k_pool_scatter_bnred): when the pool directly follows a conv block, this pass is the LAST writer of the
// block's output gradient and touches every element of it — it then also leaves the partial sums of the block's BatchNorm+ReLU backward reduce pass
// (k_bnbwd_bf16 MODE 0: g = dx masked by the ReLU, g * xhat) over the STORED (bf16-rounded) gradient: one read of yP is added, the reduce pass's reads of
// dO and yP and its launch go.  A block owns a contiguous range of cells, a thread a fixed 8-channel vector (256 % (C/8) == 0): partial sums
// [2][PB][C] for cvk_colsum_finalize, combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_pool_bwd_bnred_bf16(const __bf16* __restrict__ dout, const __bf16* __restrict__ x, PixMapH xm,
                                                            __bf16* __restrict__ dx, PixMapH dm, int accumulate, int N, int H, int W, int C,
                                                            const __bf16* __restrict__ yP, int ldp, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ part, int cpb, int PB) {
    constexpr int V = 8, U = 2;
    __shared__ float red[2][256 * V];
    const int Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2, cvn = C / V;
    const int ppp = 256 / cvn;
    const int t = threadIdx.x, cv = t % cvn, pr = t / cvn, c = cv * V;
    const long total = (long)N * Hc * Wc;
    const long cbeg = (long)blockIdx.x * cpb, cend = cbeg + cpb < total ? cbeg + cpb : total;
    const FV<V> sc = load_f32<V>(scale + c), sh = load_f32<V>(shift + c), mu = load_f32<V>(mean + c), rs = load_f32<V>(rstd + c);
    float s0[V], s1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    const u32x4 zero = {0u, 0u, 0u, 0u};
    // two cells per iteration, every load of both issued before the first use, the values kept PACKED (four registers per 8-channel vector) until
    // they are used: one cell per iteration with unpacked vectors (186 registers, a store drain between a cell's stores and the next cell's loads
    // on the one in-order counter) ran at 2.5 TB/s of its bytes and cost more than the reduce pass it replaces
    for (long cell0 = cbeg + pr; cell0 < cend; cell0 += (long)U * ppp) {
        u32x4 gw[U], xw[U][4], ow[U][4], yw[U][4];
        int64_t doff[U][4];
        bool live[U], full[U];
        unsigned inmask[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long cell = cell0 + (long)u * ppp;
            live[u] = cell < cend;
            const long cc = live[u] ? cell : cbeg;
            const int xc = (int)(cc % Wc);
            const long tq = cc / Wc;
            const int yc = (int)(tq % Hc), n = (int)(tq / Hc);
            full[u] = live[u] && yc < Ho && xc < Wo;
            gw[u] = zero;
            if (full[u]) gw[u] = *reinterpret_cast<const u32x4*>(dout + (((size_t)n * Ho + yc) * Wo + xc) * C + c);
            inmask[u] = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = 2 * yc + (k >> 1), xx = 2 * xc + (k & 1);
                const bool in = live[u] && yy < H && xx < W;
                inmask[u] |= in ? (1u << k) : 0u;
                const int m = (n * H + (in ? yy : 2 * yc)) * W + (in ? xx : 2 * xc);
                doff[u][k] = dm.off(m) + c;
                xw[u][k] = zero; ow[u][k] = zero; yw[u][k] = zero;
                if (!in) continue;
                if (full[u]) xw[u][k] = *reinterpret_cast<const u32x4*>(x + xm.off(m) + c);
                if (accumulate) ow[u][k] = *reinterpret_cast<const u32x4*>(dx + doff[u][k]);
                yw[u][k] = *reinterpret_cast<const u32x4*>(yP + (size_t)m * ldp + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            // arg-max per channel: first maximum in scan order, NaN wins (ATen's rule, as k_pool_bwd_bf16); -1: no window (odd edge)
            int cd[V];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float b0 = bf_lo(xw[u][0][i]), b1 = bf_hi(xw[u][0][i]);
                int c0 = full[u] ? 0 : -1, c1 = c0;
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float a0 = bf_lo(xw[u][k][i]), a1 = bf_hi(xw[u][k][i]);
                    if (full[u] && (a0 > b0 || a0 != a0)) { b0 = a0; c0 = k; }
                    if (full[u] && (a1 > b1 || a1 != a1)) { b1 = a1; c1 = k; }
                }
                cd[2 * i] = c0; cd[2 * i + 1] = c1;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(inmask[u] & (1u << k))) continue;
                u32x4 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r0 = cd[2 * i] == k ? bf_lo(gw[u][i]) : 0.f, r1 = cd[2 * i + 1] == k ? bf_hi(gw[u][i]) : 0.f;
                    const float o0 = accumulate ? bf_lo(ow[u][k][i]) + r0 : r0, o1 = accumulate ? bf_hi(ow[u][k][i]) + r1 : r1;
                    w[i] = bf_pack(o0, o1);
                }
                *reinterpret_cast<u32x4*>(dx + doff[u][k]) = w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {               // the sums see what was stored: the bf16-rounded gradient
                    const float st[2] = {bf_lo(w[i]), bf_hi(w[i])};
                    const float yy2[2] = {bf_lo(yw[u][k][i]), bf_hi(yw[u][k][i])};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * i + e;
                        const float gq = (yy2[e] * sc.v[j] + sh.v[j] > 0.f) ? st[e] : 0.f;
                        s0[j] += gq;
                        s1[j] += gq * ((yy2[e] - mu.v[j]) * rs.v[j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][t * V + j] = s0[j];
        red[1][t * V + j] = s1[j];
    }
    __syncthreads();
    if (t < cvn) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < ppp; ++p) {
                a += red[0][(p * cvn + t) * V + j];
                b += red[1][(p * cvn + t) * V + j];
            }
            part[(size_t)blockIdx.x * C + t * V + j] = a;
            part[(size_t)(PB + blockIdx.x) * C + t * V + j] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------ max unpool backward
// MaxUnpool2d(2) backward = a gather at the pool's arg-max: dv[cell] = dout[arg-max pixel of the cell].  The arg-max is recomputed from
// the pooled layer's stored input x exactly as in k_pool_bwd_bf16 (bf16 plans keep no index tensor: the forward unpool is that kernel
// with accumulate = 0 — a scatter of the pooled values).  Cells = floor(H/2) x floor(W/2).
template <int V>
__global__ void k_unpool_bwd_bf16(const __bf16* __restrict__ dout, const __bf16* __restrict__ x, PixMapH xm, __bf16* __restrict__ dv,
                                  int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, cvn = C / V;
    const long total = (long)N * Ho * Wo * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xc = (int)(t % Wo);
        t /= Wo;
        const int yc = (int)(t % Ho), n = (int)(t / Ho);
        const int c = cv * V;
        FV<V> best, g[4], o;
        int cd[V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = (n * H + 2 * yc + (k >> 1)) * W + 2 * xc + (k & 1);
            const FV<V> v = load_bf16<V>(x + xm.off(m) + c);
            g[k] = load_bf16<V>(dout + (size_t)m * C + c);
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (k == 0 || v.v[j] > best.v[j] || v.v[j] != v.v[j]) { best.v[j] = v.v[j]; cd[j] = k; }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = cd[j] == 0 ? g[0].v[j] : (cd[j] == 1 ? g[1].v[j] : (cd[j] == 2 ? g[2].v[j] : g[3].v[j]));
        store_bf16<V>(dv + (((size_t)n * Ho + yc) * Wo + xc) * C + c, o);
    }
}

// ------------------------------------------------------------------------------------------------ bilinear x2
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap make_tap(int dst, float scale, int n_in) {   // ATen align_corners source index (pointwise.hip)
    Tap t;
    const float src = scale * (float)dst;
    t.i0 = (int)src;
    if (t.i0 > n_in - 1) t.i0 = n_in - 1;
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

template <int V>
__global__ void k_bilinear_fwd_bf16(const __bf16* __restrict__ x, __bf16* __restrict__ out, int N, int H, int W, int C, float sy, float sx) {
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const long total = (long)N * Ho * Wo * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho), n = (int)(t / Ho);
        const Tap ty = make_tap(yo, sy, H), tx = make_tap(xo, sx, W);
        const __bf16* b = x + ((size_t)n * H * W) * C + cv * V;
        const FV<V> v00 = load_bf16<V>(b + ((size_t)ty.i0 * W + tx.i0) * C), v01 = load_bf16<V>(b + ((size_t)ty.i0 * W + tx.i1) * C);
        const FV<V> v10 = load_bf16<V>(b + ((size_t)ty.i1 * W + tx.i0) * C), v11 = load_bf16<V>(b + ((size_t)ty.i1 * W + tx.i1) * C);
        FV<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j)
            o.v[j] = ty.l0 * (tx.l0 * v00.v[j] + tx.l1 * v01.v[j]) + ty.l1 * (tx.l0 * v10.v[j] + tx.l1 * v11.v[j]);
        store_bf16<V>(out + i * V, o);
    }
}

// LDS-tiled forward for 128-channel chunks (pointwise.hip k_bilinear_fwd_tiled has the reasoning): a block = 4 x 32 output
// pixels x 128 channels; the <= 4 x 18 input pixels of the tile are loaded once as raw 16-byte vectors, the taps read LDS.
// Same float operations as the flat kernel.
__global__ __launch_bounds__(256) void k_bilinear_fwd_bf16_tiled(const __bf16* __restrict__ x, __bf16* __restrict__ out, int H, int W, int C,
                                                                float sy, float sx, int tilesX) {
    constexpr int TY = 4, TX = 32, IY = 4, IX = 18;
    __shared__ __attribute__((aligned(16))) __bf16 tile[IY * IX * 16 * 8];
    const int Ho = 2 * H, Wo = 2 * W;
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX, c0 = blockIdx.y * 128, n = blockIdx.z;
    const int yo0 = ty * TY, xo0 = tx * TX;
    const int iy0 = make_tap(yo0, sy, H).i0, ix0 = make_tap(xo0, sx, W).i0;
    const int t = threadIdx.x, cv = t & 15;
    const __bf16* const xb = x + ((size_t)n * H * W) * C + c0 + cv * 8;
    for (int p = t >> 4; p < IY * IX; p += 16) {
        const int ry = p / IX, rx = p - ry * IX;
        const int iy = min(iy0 + ry, H - 1), ix = min(ix0 + rx, W - 1);
        *reinterpret_cast<u32x4*>(tile + (p * 16 + cv) * 8) = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * W + ix) * C);
    }
    __syncthreads();
    for (int p = t >> 4; p < TY * TX; p += 16) {
        const int oy = p / TX, ox = p - oy * TX;
        const int yo = yo0 + oy, xo = xo0 + ox;
        if (yo >= Ho || xo >= Wo) continue;
        const Tap ty_ = make_tap(yo, sy, H), tx_ = make_tap(xo, sx, W);
        const int r0 = (ty_.i0 - iy0) * IX, r1 = (ty_.i1 - iy0) * IX, q0 = tx_.i0 - ix0, q1 = tx_.i1 - ix0;
        const FV<8> v00 = load_bf16<8>(tile + ((r0 + q0) * 16 + cv) * 8), v01 = load_bf16<8>(tile + ((r0 + q1) * 16 + cv) * 8);
        const FV<8> v10 = load_bf16<8>(tile + ((r1 + q0) * 16 + cv) * 8), v11 = load_bf16<8>(tile + ((r1 + q1) * 16 + cv) * 8);
        FV<8> o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o.v[j] = ty_.l0 * (tx_.l0 * v00.v[j] + tx_.l1 * v01.v[j]) + ty_.l1 * (tx_.l0 * v10.v[j] + tx_.l1 * v11.v[j]);
        store_bf16<8>(out + (((size_t)n * Ho + yo) * Wo + xo) * C + c0 + cv * 8, o);
    }
}

// backward = gather over the output pixels whose taps touch input pixel (yi, xi), as pointwise.hip k_bilinear_bwd: a fixed 6 x 6
// candidate window with per-candidate weights (all contributing loads in flight together; the first version walked data-dependent
// loop bounds with one load in flight), one workgroup row = one input row, rows dealt to the XCDs in contiguous chunks (neighbouring
// input rows share two of their ~four gradient rows: they stay in one L2).
template <int V>
__global__ __launch_bounds__(256) void k_bilinear_bwd_bf16(const __bf16* __restrict__ dout, __bf16* __restrict__ dx, int N, int H, int W, int C,
                                                          float sy, float sx, int gx) {
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const unsigned lid = (unsigned)cvk_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned row = lid / (unsigned)gx, bx = lid - row * gx;
    const int n = (int)(row / (unsigned)H), yi = (int)(row - (unsigned)n * H);
    const unsigned idx = bx * 256 + threadIdx.x;
    if (idx >= (unsigned)(W * cvn)) return;
    const int xi = (int)(idx / (unsigned)cvn), cv = (int)idx - xi * cvn;
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
    FV<V> acc;
#pragma unroll
    for (int j = 0; j < V; ++j) acc.v[j] = 0.f;
    const __bf16* b = dout + ((size_t)n * Ho * Wo) * C + cv * V;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {                             // block-uniform
        if (wy[ky] == 0.f) continue;
        const int yo = 2 * yi - 2 + ky;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            if (wx[kx] == 0.f) continue;
            const int xo = 2 * xi - 2 + kx;
            const FV<V> g = load_bf16<V>(b + ((size_t)yo * Wo + xo) * C);
            const float w = wy[ky] * wx[kx];
#pragma unroll
            for (int j = 0; j < V; ++j) acc.v[j] += w * g.v[j];
        }
    }
    store_bf16<V>(dx + (((size_t)n * H + yi) * W) * C + (size_t)idx * V, acc);
}

__global__ void k_zero_frame_bf16(cvk_viewh b, int N, int H, int W, int C, int y0, int x0, int h, int w) {
    const int gv = C / 4;
    const long total = (long)N * H * W * gv;
    __bf16* base = reinterpret_cast<__bf16*>(b.ptr);
    CVK_GRID_STRIDE(i, total) {
        const int g = (int)(i % gv);
        long t = i / gv;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H), n = (int)(t / H);
        const bool inside = y >= y0 && y < y0 + h && x >= x0 && x < x0 + w;
        if (!inside) *reinterpret_cast<u32x2*>(base + n * b.sN + y * b.sY + x * b.sX + 4 * g) = u32x2{0u, 0u};
    }
}

inline bool viewok(const cvk_viewh& v, int V, bool f32) {
    const uintptr_t al = f32 ? 16 : (V == 8 ? 16 : 8);
    return v.ptr && (((uintptr_t)v.ptr) % al) == 0 && ((v.sN | v.sY | v.sX) % (f32 ? 4 : V)) == 0;
}

int bwd_blocks(int M) {
    int pb = cvk_cdiv(M, 16);
    pb = pb < 1024 ? pb : 1024;
    const int rows = cvk_cdiv(M, pb);
    return cvk_cdiv(M, rows);
}

}  // namespace

extern "C" int cvk_import_nchw_bf16(const float* src, int64_t sN, int64_t sC, int64_t sH, int64_t sW, void* dst, int ld, int N, int C,
                                    int H, int W, void* stream) {
    CVK_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C && ld % 8 == 0 && cvk_aligned16(dst), "cvk_import_nchw_bf16: bad arguments");
    hipLaunchKernelGGL(k_import_bf16, dim3(grid_for((long)N * H * W * (ld / 8))), dim3(256), 0, (hipStream_t)stream, src, sN, sC, sH, sW,
                       (__bf16*)dst, ld, N, C, H, W);
    CVK_LAUNCH_RETURN("cvk_import_nchw_bf16");
}

extern "C" int cvk_bn_relu_apply_bf16(const void* y, int ldy, const float* scale, const float* shift, cvk_viewh out, int out_f32,
                                      void* pool, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(y && scale && shift && out.ptr, "cvk_bn_relu_apply_bf16: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && (long)N * H * W < (1L << 31), "cvk_bn_relu_apply_bf16: bad shape");
    CVK_CHECK_ARG(C % 4 == 0 && ldy % 4 == 0 && cvk_aligned16(scale) && cvk_aligned16(shift), "cvk_bn_relu_apply_bf16: C and ldy must be multiples of 4");
    CVK_CHECK_ARG(!(pool && out_f32), "cvk_bn_relu_apply_bf16: the fused pool writes bf16 only");
    CVK_CHECK_ARG(!pool || (H >= 2 && W >= 2), "cvk_bn_relu_apply_bf16: max pool needs H, W >= 2");
    const PixMapH om = make_map(out, H, W);
    hipStream_t s = (hipStream_t)stream;
    const bool v8 = C % 8 == 0 && ldy % 8 == 0 && cvk_aligned16(y) && viewok(out, 8, out_f32 != 0) && (!pool || cvk_aligned16(pool));
    CVK_CHECK_ARG(v8 || viewok(out, 4, out_f32 != 0), "cvk_bn_relu_apply_bf16: misaligned output view");
    const long cells = pool ? (long)N * ((H + 1) / 2) * ((W + 1) / 2) : (long)N * H * W;
#define CVK_AP(V_, F_, P_) hipLaunchKernelGGL((k_apply_bf16<V_, F_, P_>), dim3(grid_for(cells * (C / V_))), dim3(256), 0, s, (const __bf16*)y, ldy, scale, shift, out.ptr, om, (__bf16*)pool, N, H, W, C)
    const int nt = cvk_knob("CVK_STREAM_HINTS", 5);      // experiments build: 0 = no streaming hints (A/B timing)
    if (v8 && nt >= 2 && !out_f32) {
        if (pool) hipLaunchKernelGGL((k_apply_bf16<8, false, true, true>), dim3(grid_for(cells * (C / 8))), dim3(256), 0, s, (const __bf16*)y, ldy, scale, shift, out.ptr, om, (__bf16*)pool, N, H, W, C);
        else hipLaunchKernelGGL((k_apply_bf16<8, false, false, true>), dim3(grid_for(cells * (C / 8))), dim3(256), 0, s, (const __bf16*)y, ldy, scale, shift, out.ptr, om, (__bf16*)pool, N, H, W, C);
    } else if (v8) {
        if (pool) CVK_AP(8, false, true);
        else if (out_f32) CVK_AP(8, true, false);
        else CVK_AP(8, false, false);
    } else {
        if (pool) CVK_AP(4, false, true);
        else if (out_f32) CVK_AP(4, true, false);
        else CVK_AP(4, false, false);
    }
#undef CVK_AP
    CVK_LAUNCH_RETURN("cvk_bn_relu_apply_bf16");
}

extern "C" int cvk_bn_bwd_blocks_bf16(int M) { return M > 0 ? bwd_blocks(M) : 0; }

static int bnbwd_launch(int mode, cvk_viewh dout, int dout_f32, const void* y, int ldy, const float* scale, const float* shift,
                        const float* mean, const float* rstd, const float* dgamma, const float* dbeta, void* dy, int ld_dy, float* part,
                        int N, int H, int W, int C, int use_batch_stats, void* stream, const char* name) {
    CVK_CHECK_ARG(dout.ptr && y && scale && shift && mean && rstd, "%s: null pointer", name);
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && (long)N * H * W < (1L << 31), "%s: bad shape", name);
    CVK_CHECK_ARG(C % 4 == 0 && ldy % 4 == 0 && C <= 2048, "%s: C must be a multiple of 4 and <= 2048", name);
    const int M = N * H * W;
    const PixMapH dm = make_map(dout, H, W);
    const int PB = bwd_blocks(M), rows = cvk_cdiv(M, PB);
    hipStream_t s = (hipStream_t)stream;
    const bool v8 = C % 8 == 0 && ldy % 8 == 0 && cvk_aligned16(y) && viewok(dout, 8, dout_f32 != 0) && (mode == 0 || (ld_dy % 8 == 0 && cvk_aligned16(dy)));
    CVK_CHECK_ARG(v8 || viewok(dout, 4, dout_f32 != 0), "%s: misaligned gradient view", name);
    // a block's 256 threads own the C / V channel vectors (pixels-per-pass = 256 / (C / V)): more than 256 vectors would leave
    // every thread idle and the partial sums unwritten
    CVK_CHECK_ARG(C / (v8 ? 8 : 4) <= 256, "%s: C=%d needs more than 256 channel vectors of %d (C <= 1024 for 4-wide, 2048 for 8-wide access)", name, C, v8 ? 8 : 4);
#define CVK_BB(V_, M_, F_) hipLaunchKernelGGL((k_bnbwd_bf16<V_, M_, F_>), dim3(PB), dim3(256), 0, s, dout.ptr, dm, (const __bf16*)y, ldy, scale, shift, mean, rstd, dgamma, dbeta, (__bf16*)dy, ld_dy, part, M, C, rows, PB, use_batch_stats)
    const int nt = cvk_knob("CVK_STREAM_HINTS", 5);      // experiments build: 0 = no streaming hints (A/B timing)
    if (v8) {
        if (mode == 0 && nt >= 4 && !dout_f32 && (size_t)M * C * 2 >= ((size_t)128 << 20)) hipLaunchKernelGGL((k_bnbwd_bf16<8, 0, false, true>), dim3(PB), dim3(256), 0, s, dout.ptr, dm, (const __bf16*)y, ldy, scale, shift, mean, rstd, dgamma, dbeta, (__bf16*)dy, ld_dy, part, M, C, rows, PB, use_batch_stats);
        else if (mode == 0) { if (dout_f32) CVK_BB(8, 0, true); else CVK_BB(8, 0, false); }
        else if (nt && !dout_f32 && (nt < 3 || (size_t)M * C * 2 >= ((size_t)128 << 20))) hipLaunchKernelGGL((k_bnbwd_bf16<8, 1, false, true>), dim3(PB), dim3(256), 0, s, dout.ptr, dm, (const __bf16*)y, ldy, scale, shift, mean, rstd, dgamma, dbeta, (__bf16*)dy, ld_dy, part, M, C, rows, PB, use_batch_stats);
        else           { if (dout_f32) CVK_BB(8, 1, true); else CVK_BB(8, 1, false); }
    } else {
        if (mode == 0) { if (dout_f32) CVK_BB(4, 0, true); else CVK_BB(4, 0, false); }
        else           { if (dout_f32) CVK_BB(4, 1, true); else CVK_BB(4, 1, false); }
    }
#undef CVK_BB
    CVK_LAUNCH_RETURN(name);
}

extern "C" int cvk_bn_bwd_reduce_bf16(cvk_viewh dout, int dout_f32, const void* y, int ldy, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, float* part, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(part, "cvk_bn_bwd_reduce_bf16: null partial buffer");
    return bnbwd_launch(0, dout, dout_f32, y, ldy, scale, shift, mean, rstd, nullptr, nullptr, nullptr, 0, part, N, H, W, C, 1, stream,
                        "cvk_bn_bwd_reduce_bf16");
}

extern "C" int cvk_bn_bwd_dx_bf16(cvk_viewh dout, int dout_f32, const void* y, int ldy, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, const float* dgamma, const float* dbeta, void* dy, int ld_dy,
                                  float* dbias_part, int N, int H, int W, int C, int use_batch_stats, void* stream) {
    CVK_CHECK_ARG(dy && ld_dy >= C && ld_dy % 4 == 0 && (((uintptr_t)dy) & 7u) == 0, "cvk_bn_bwd_dx_bf16: bad dy");
    CVK_CHECK_ARG(!use_batch_stats || (dgamma && dbeta), "cvk_bn_bwd_dx_bf16: dgamma/dbeta required in training mode");
    return bnbwd_launch(1, dout, dout_f32, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, dbias_part, N, H, W, C,
                        use_batch_stats, stream, "cvk_bn_bwd_dx_bf16");
}

extern "C" int cvk_maxpool2x2_bwd_bf16(const void* dout, cvk_viewh x, cvk_viewh dx, int accumulate, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && x.ptr && dx.ptr && N > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, "cvk_maxpool2x2_bwd_bf16: bad arguments (C % 8)");
    CVK_CHECK_ARG(cvk_aligned16(dout) && viewok(x, 8, false) && viewok(dx, 8, false), "cvk_maxpool2x2_bwd_bf16: misaligned view");
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    hipLaunchKernelGGL(k_pool_bwd_bf16<8>, dim3(grid_for(cells * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dout,
                       (const __bf16*)x.ptr, make_map(x, H, W), (__bf16*)dx.ptr, make_map(dx, H, W), accumulate, N, H, W, C);
    CVK_LAUNCH_RETURN("cvk_maxpool2x2_bwd_bf16");
}

// blocks (= partial sums per channel) of cvk_maxpool2x2_bwd_bnred_bf16; 0: the shape is not supported (C/8 must divide 256)
extern "C" int cvk_maxpool2x2_bwd_bnred_blocks_bf16(int N, int H, int W, int C) {
    if (N <= 0 || H < 2 || W < 2 || C < 8 || C % 8 || C > 2048 || 256 % (C / 8)) return 0;
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const int ppp = 256 / (C / 8);
    const long want = (cells + 4L * ppp - 1) / (4L * ppp);          // >= 4 cells per thread
    return (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

extern "C" int cvk_maxpool2x2_bwd_bnred_bf16(const void* dout, cvk_viewh x, cvk_viewh dx, int accumulate, int N, int H, int W, int C, const void* yP,
                                             int ldp, const float* scale, const float* shift, const float* mean, const float* rstd, float* part,
                                             void* stream) {
    CVK_CHECK_ARG(dout && x.ptr && dx.ptr && yP && scale && shift && mean && rstd && part, "cvk_maxpool2x2_bwd_bnred_bf16: null pointer");
    const int PB = cvk_maxpool2x2_bwd_bnred_blocks_bf16(N, H, W, C);
    CVK_CHECK_ARG(PB > 0 && ldp >= C && ldp % 8 == 0, "cvk_maxpool2x2_bwd_bnred_bf16: unsupported shape (C=%d: C/8 must divide 256)", C);
    CVK_CHECK_ARG(cvk_aligned16(dout) && viewok(x, 8, false) && viewok(dx, 8, false) && cvk_aligned16(yP) && cvk_aligned16(scale) && cvk_aligned16(shift) &&
                  cvk_aligned16(mean) && cvk_aligned16(rstd), "cvk_maxpool2x2_bwd_bnred_bf16: misaligned pointer or view");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31), "cvk_maxpool2x2_bwd_bnred_bf16: more than 2^31 pixels");
    const long cells = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const int cpb = (int)((cells + PB - 1) / PB);
    hipLaunchKernelGGL(k_pool_bwd_bnred_bf16, dim3(PB), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dout, (const __bf16*)x.ptr, make_map(x, H, W),
                       (__bf16*)dx.ptr, make_map(dx, H, W), accumulate, N, H, W, C, (const __bf16*)yP, ldp, scale, shift, mean, rstd, part, cpb, PB);
    CVK_LAUNCH_RETURN("cvk_maxpool2x2_bwd_bnred_bf16");
}

extern "C" int cvk_maxunpool2x2_bwd_bf16(const void* dout, cvk_viewh x, void* dv, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && x.ptr && dv && N > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, "cvk_maxunpool2x2_bwd_bf16: bad arguments (C % 8)");
    CVK_CHECK_ARG(cvk_aligned16(dout) && cvk_aligned16(dv) && viewok(x, 8, false), "cvk_maxunpool2x2_bwd_bf16: misaligned pointer or view");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31), "cvk_maxunpool2x2_bwd_bf16: more than 2^31 pixels");
    const long cells = (long)N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(k_unpool_bwd_bf16<8>, dim3(grid_for(cells * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dout,
                       (const __bf16*)x.ptr, make_map(x, H, W), (__bf16*)dv, N, H, W, C);
    CVK_LAUNCH_RETURN("cvk_maxunpool2x2_bwd_bf16");
}

static inline float ac_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

extern "C" int cvk_bilinear_up2_fwd_bf16(const void* x, void* out, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && cvk_aligned16(x) && cvk_aligned16(out), "cvk_bilinear_up2_fwd_bf16: bad arguments");
    if (C % 128 == 0 && H >= 2 && W >= 2 && N <= 65535 && C / 128 <= 65535) {
        const int tilesX = cvk_cdiv(2 * W, 32), tilesY = cvk_cdiv(2 * H, 4);
        hipLaunchKernelGGL(k_bilinear_fwd_bf16_tiled, dim3(tilesX * tilesY, C / 128, N), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)x, (__bf16*)out, H, W, C, ac_scale(H, 2 * H), ac_scale(W, 2 * W), tilesX);
        CVK_LAUNCH_RETURN("cvk_bilinear_up2_fwd_bf16");
    }
    hipLaunchKernelGGL(k_bilinear_fwd_bf16<8>, dim3(grid_for((long)N * 4 * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x, (__bf16*)out, N, H, W, C, ac_scale(H, 2 * H), ac_scale(W, 2 * W));
    CVK_LAUNCH_RETURN("cvk_bilinear_up2_fwd_bf16");
}

extern "C" int cvk_bilinear_up2_bwd_bf16(const void* dout, void* dx, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && cvk_aligned16(dout) && cvk_aligned16(dx), "cvk_bilinear_up2_bwd_bf16: bad arguments");
    const int gx = cvk_cdiv((long)W * (C / 8), 256);
    CVK_CHECK_ARG((long)gx * H * N < (1L << 31) - 8, "cvk_bilinear_up2_bwd_bf16: too many workgroups");
    hipLaunchKernelGGL(k_bilinear_bwd_bf16<8>, dim3((unsigned)((long)gx * H * N)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dout, (__bf16*)dx, N, H, W, C, ac_scale(H, 2 * H), ac_scale(W, 2 * W), gx);
    CVK_LAUNCH_RETURN("cvk_bilinear_up2_bwd_bf16");
}

extern "C" int cvk_zero_frame_bf16(cvk_viewh buf, int N, int H, int W, int C, int y0, int x0, int h, int w, void* stream) {
    CVK_CHECK_ARG(buf.ptr && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W, "cvk_zero_frame_bf16: bad arguments");
    CVK_CHECK_ARG(viewok(buf, 4, false), "cvk_zero_frame_bf16: misaligned view");
    hipLaunchKernelGGL(k_zero_frame_bf16, dim3(grid_for((long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, buf, N, H, W, C, y0, x0, h, w);
    CVK_LAUNCH_RETURN("cvk_zero_frame_bf16");
}
