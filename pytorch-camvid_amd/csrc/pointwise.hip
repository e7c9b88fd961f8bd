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

template <int V>
__global__ void k_bilinear_fwd(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C, float sy,
                               float sx) {
    typedef typename VT_<V>::T VT;
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const long total = (long)N * Ho * Wo * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho), n = (int)(t / Ho);
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
        *reinterpret_cast<VT*>(out + i * V) = o;
    }
}

// gather form of the transpose: input pixel (y,x) sums w_y(yo)*w_x(xo)*dout[yo][xo] over the <= 6x6 output window
// whose taps can touch it; weights are recomputed with exactly the forward arithmetic.
template <int V>
__global__ void k_bilinear_bwd(const float* __restrict__ dout, float* __restrict__ dx, int N, int H, int W, int C, float sy,
                               float sx) {
    typedef typename VT_<V>::T VT;
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const long total = (long)N * H * W * cvn;
    CVK_GRID_STRIDE(i, total) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xi = (int)(t % W);
        t /= W;
        const int yi = (int)(t % H), n = (int)(t / H);
        // candidate output rows/cols: src in (yi-1, yi+1)  =>  dst in ((yi-1)/s, (yi+1)/s)
        int ylo = 0, yhi = Ho - 1, xlo = 0, xhi = Wo - 1;
        if (sy > 0.f) { ylo = max(0, (int)floorf((float)(yi - 1) / sy)); yhi = min(Ho - 1, (int)ceilf((float)(yi + 1) / sy)); }
        if (sx > 0.f) { xlo = max(0, (int)floorf((float)(xi - 1) / sx)); xhi = min(Wo - 1, (int)ceilf((float)(xi + 1) / sx)); }
        VT acc;
#pragma unroll
        for (int j = 0; j < V; ++j) el<V>(acc, j) = 0.f;
        const float* b = dout + ((long)n * Ho * Wo) * C + cv * V;
        for (int yo = ylo; yo <= yhi; ++yo) {
            const Tap ty = make_tap(yo, sy, H);
            const float wy = (ty.i0 == yi ? ty.l0 : 0.f) + (ty.i1 == yi ? ty.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xlo; xo <= xhi; ++xo) {
                const Tap tx = make_tap(xo, sx, W);
                const float wx = (tx.i0 == xi ? tx.l0 : 0.f) + (tx.i1 == xi ? tx.l1 : 0.f);
                if (wx == 0.f) continue;
                VT g = *reinterpret_cast<const VT*>(b + ((long)yo * Wo + xo) * C);
                const float w = wy * wx;
#pragma unroll
                for (int j = 0; j < V; ++j) el<V>(acc, j) += w * el<V>(g, j);
            }
        }
        *reinterpret_cast<VT*>(dx + i * V) = acc;
    }
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
    if (v4ok(C, x, out))
        hipLaunchKernelGGL(k_bilinear_fwd<4>, dim3(grid_for((long)N * 4 * H * W * (C / 4))), dim3(256), 0, s, x, out, N, H, W, C, sy, sx);
    else
        hipLaunchKernelGGL(k_bilinear_fwd<1>, dim3(grid_for((long)N * 4 * H * W * C)), dim3(256), 0, s, x, out, N, H, W, C, sy, sx);
    CVK_LAUNCH_RETURN("cvk_bilinear_up2_fwd");
}

extern "C" int cvk_bilinear_up2_bwd(const float* dout, float* dx, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(dout && dx && N > 0 && H > 0 && W > 0 && C > 0, "cvk_bilinear_up2_bwd: bad arguments");
    const float sy = ac_scale(H, 2 * H), sx = ac_scale(W, 2 * W);
    hipStream_t s = (hipStream_t)stream;
    if (v4ok(C, dout, dx))
        hipLaunchKernelGGL(k_bilinear_bwd<4>, dim3(grid_for((long)N * H * W * (C / 4))), dim3(256), 0, s, dout, dx, N, H, W, C, sy, sx);
    else
        hipLaunchKernelGGL(k_bilinear_bwd<1>, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, s, dout, dx, N, H, W, C, sy, sx);
    CVK_LAUNCH_RETURN("cvk_bilinear_up2_bwd");
}
