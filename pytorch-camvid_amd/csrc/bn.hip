// bn.hip — BatchNorm2d (+ fused ReLU) forward/backward for NHWC fp32, gfx950.
//
// Replaces ATen native_batch_norm / native_batch_norm_backward / relu_ / threshold_backward behind
// nn.BatchNorm2d + nn.ReLU(inplace=True) of the reference (models/unet.py:12-13, models/segnet.py:9-10).
// All kernels are HBM-bound streams: 16-byte vector access along the channel (innermost) dimension, every tensor
// read at most once per pass.  Per-channel reductions: each thread owns a fixed 4-channel vector and walks pixels,
// the block combines through LDS, and the few hundred block partials are summed in fp64 in a fixed order
// (bitwise reproducible; no float atomics).
//   forward statistics arrive already reduced to 64-row partials from the conv epilogue (conv3x3.hip), so the
//   activation tensor is NOT re-read for the mean/variance: training forward = 1 read + 1 write.
#include "cvk_common.h"

namespace {

// ---------------------------------------------------------------------------------------------- forward statistics
// level 1: [2][P][C] float partials (sum, M2 about the partial mean, n_p = min(64, M-64p) rows) ->
//          [G][C][3] double: (sum s_p, sum q_p, sum s_p^2/n_p)
__global__ __launch_bounds__(256) void k_bn_stats_l1(const float* __restrict__ stats, double* __restrict__ ws, int P, int M,
                                                    int C, int rows_per_g) {
    __shared__ double red[3][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int pbeg = blockIdx.y * rows_per_g, pend = min(P, pbeg + rows_per_g);
    double a = 0.0, b = 0.0, q = 0.0;
    if (c < C) {
        // every granule holds CVK_STAT_ROWS rows except the last one of the tensor: 1/n is a constant (exact power of two)
        // for all but that one — no fp64 division in the loop; two independent chains hide the load latency
        const double inv_full = 1.0 / (double)CVK_STAT_ROWS;
        const int last = P - 1;
        const double inv_last = 1.0 / (double)(M - last * CVK_STAT_ROWS);
        double a1 = 0.0, b1 = 0.0, q1 = 0.0;
        int p = pbeg + g;
        for (; p + 4 < pend; p += 8) {
            const double s0 = stats[(size_t)p * C + c], s1 = stats[(size_t)(p + 4) * C + c];
            const double m0 = stats[(size_t)(P + p) * C + c], m1 = stats[(size_t)(P + p + 4) * C + c];
            a += s0; b += m0; q += s0 * s0 * (p == last ? inv_last : inv_full);
            a1 += s1; b1 += m1; q1 += s1 * s1 * (p + 4 == last ? inv_last : inv_full);
        }
        for (; p < pend; p += 4) {
            const double s0 = stats[(size_t)p * C + c];
            a += s0; b += stats[(size_t)(P + p) * C + c]; q += s0 * s0 * (p == last ? inv_last : inv_full);
        }
        a += a1; b += b1; q += q1;
    }
    red[0][g][threadIdx.x & 63] = a;
    red[1][g][threadIdx.x & 63] = b;
    red[2][g][threadIdx.x & 63] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        const int l = threadIdx.x;
        double* o = ws + ((size_t)blockIdx.y * C + c) * 3;
        o[0] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        o[1] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
        o[2] = (red[2][0][l] + red[2][1][l]) + (red[2][2][l] + red[2][3][l]);
    }
}

// level 1 for partials of ARBITRARY size (the bf16-storage convolution reduces per 8 x 32 pixel tile, ragged at the frame
// border): cnt[p] = number of pixels behind partial p.  Same outputs as k_bn_stats_l1.
__global__ __launch_bounds__(256) void k_bn_stats_l1_counts(const float* __restrict__ stats, const float* __restrict__ cnt,
                                                           double* __restrict__ ws, int P, int C, int rows_per_g) {
    __shared__ double red[3][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int pbeg = blockIdx.y * rows_per_g, pend = min(P, pbeg + rows_per_g);
    double a = 0.0, b = 0.0, q = 0.0;
    if (c < C)
        for (int p = pbeg + g; p < pend; p += 4) {
            const double s0 = stats[(size_t)p * C + c];
            a += s0;
            b += stats[(size_t)(P + p) * C + c];
            q += s0 * s0 / (double)cnt[p];
        }
    red[0][g][threadIdx.x & 63] = a;
    red[1][g][threadIdx.x & 63] = b;
    red[2][g][threadIdx.x & 63] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        const int l = threadIdx.x;
        double* o = ws + ((size_t)blockIdx.y * C + c) * 3;
        o[0] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        o[1] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
        o[2] = (red[2][0][l] + red[2][1][l]) + (red[2][2][l] + red[2][3][l]);
    }
}

// level 2: one thread per channel; M2 = sum q_p + sum s_p^2/n_p - S^2/M (Chan, evaluated in fp64)
__global__ __launch_bounds__(256) void k_bn_stats_l2(const double* __restrict__ ws, int G, int M, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* mean, float* rstd, float* scale, float* shift,
                              float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps) {
    __shared__ double red[3][4][64];
    const int l = threadIdx.x & 63, lane = threadIdx.x >> 6;      // 64 channels x 4 lanes over the G partial rows
    const int c = blockIdx.x * 64 + l;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
    double a = 0.0, b = 0.0, q = 0.0;
    if (c < C)
        for (int g = lane; g < G; g += 4) {
            const double* o = ws + ((size_t)g * C + c) * 3;
            a += o[0];
            b += o[1];
            q += o[2];
        }
    red[0][lane][l] = a;
    red[1][lane][l] = b;
    red[2][lane][l] = q;
    __syncthreads();
    if (lane != 0 || c >= C) return;
    a = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    b = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    q = (red[2][0][l] + red[2][1][l]) + (red[2][2][l] + red[2][3][l]);
    const double mu = a / (double)M;
    double m2 = b + q - a * a / (double)M;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / (double)M;  // biased, used for normalisation
    const double rs = 1.0 / sqrt(var + (double)eps);
    const float fmu = (float)mu, frs = (float)rs;
    mean[c] = fmu;
    rstd[c] = frs;
    const float sc = gamma[c] * frs;
    scale[c] = sc;
    shift[c] = beta[c] - fmu * sc;
    if (running_mean != nullptr) {
        const double unb = M > 1 ? m2 / (double)(M - 1) : var;  // running_var tracks the unbiased estimate
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fmu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void k_bn_eval_params(const float* gamma, const float* beta, const float* rm, const float* rv, float* mean,
                                 float* rstd, float* scale, float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rs = 1.f / sqrtf(rv[c] + eps);
    mean[c] = rm[c];
    rstd[c] = rs;
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---------------------------------------------------------------------------------------------- view addressing
struct PixMap {  // pixel m (row-major over N,H,W) -> float offset inside a strided view
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

PixMap make_map(const cvk_view& v, int H, int W) {
    PixMap p;
    p.sN = v.sN; p.sY = v.sY; p.sX = v.sX; p.H = H; p.W = W;
    p.linear = (v.sY == (int64_t)W * v.sX && v.sN == (int64_t)H * v.sY) ? 1 : 0;
    return p;
}

template <int V> struct Vec;
template <> struct Vec<4> { typedef f32x4 T; };
template <> struct Vec<1> { typedef float T; };
template <int V> __device__ __forceinline__ float vget(const typename Vec<V>::T& v, int j);
template <> __device__ __forceinline__ float vget<4>(const f32x4& v, int j) { return v[j]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <int V> __device__ __forceinline__ void vset(typename Vec<V>::T& v, int j, float x);
template <> __device__ __forceinline__ void vset<4>(f32x4& v, int j, float x) { v[j] = x; }
template <> __device__ __forceinline__ void vset<1>(float& v, int, float x) { v = x; }

// ---------------------------------------------------------------------------------------------- apply + ReLU
template <int V>
__global__ __launch_bounds__(256) void k_bn_relu_apply(const float* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ out, PixMap om,
                                                      int M, int C, unsigned* __restrict__ amax) {
    typedef typename Vec<V>::T VT;
    const int cvn = C / V;  // vectors per pixel
    const long total = (long)M * cvn;
    float mx = 0.f;         // the largest activation this thread wrote (>= 0 behind the ReLU)
    const unsigned snap = cvk_amax_snapshot(amax);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / cvn);
        const int c = (int)(i - (long)m * cvn) * V;
        const VT v = *reinterpret_cast<const VT*>(y + (size_t)m * ldy + c);
        const VT sc = *reinterpret_cast<const VT*>(scale + c);
        const VT sh = *reinterpret_cast<const VT*>(shift + c);
        VT o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float r = fmaxf(vget<V>(v, j) * vget<V>(sc, j) + vget<V>(sh, j), 0.f);
            vset<V>(o, j, r);
            mx = fmaxf(mx, r);
        }
        *reinterpret_cast<VT*>(out + om.off(m) + c) = o;
    }
    if (amax != nullptr) cvk_amax_publish_wg(mx, amax, snap);    // consumed by the opt-in fp16 split-operand transforms (csrc/split_fmt.h)
}

// BN-apply + ReLU with the 2x2 max pool of the result fused in (nn.MaxPool2d(2,2) directly behind a conv block:
// /root/reference/models/unet.py:100-109, models/segnet.py:79): one thread = one 2x2 cell x 4 channels; writes the four
// activations through the (concat) view and, for whole windows, the pooled value (+ the arg-max code 0..3 in window scan order,
// first maximum, NaN propagating — exactly k_maxpool_fwd of pointwise.hip).  Saves the pool pass's re-read of the activation.
__global__ __launch_bounds__(256) void k_bn_relu_apply_pool(const float* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ out, PixMap om,
                                                           float* __restrict__ pool, unsigned char* __restrict__ code, int N, int H,
                                                           int W, int C, unsigned* __restrict__ amax_out, unsigned* __restrict__ amax_pool) {
    const int cvn = C >> 2, Hc = (H + 1) >> 1, Wc = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Hc * Wc * cvn;
    float mxo = 0.f, mxp = 0.f;         // largest activation / largest pooled value this thread wrote
    const unsigned snapo = cvk_amax_snapshot(amax_out), snapp = cvk_amax_snapshot(amax_pool);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        long t = i / cvn;
        const int xc = (int)(t % Wc);
        t /= Wc;
        const int yc = (int)(t % Hc), n = (int)(t / Hc);
        const int c = cv * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = 2 * yc + (k >> 1), xx = 2 * xc + (k & 1);
            if (yy < H && xx < W) {
                const int m = (n * H + yy) * W + xx;
                const f32x4 a = *reinterpret_cast<const f32x4*>(y + (size_t)m * ldy + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[k][j] = fmaxf(a[j] * sc[j] + sh[j], 0.f);
                    mxo = fmaxf(mxo, v[k][j]);
                }
                *reinterpret_cast<f32x4*>(out + om.off(m) + c) = v[k];
            }
        }
        if (yc < Ho && xc < Wo) {
            f32x4 best = v[0];
            unsigned char cd[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = v[k][j], b = best[j];
                    if (a > b || a != a) { best[j] = a; cd[j] = (unsigned char)k; }
                }
            const long o = (((long)n * Ho + yc) * Wo + xc) * C + c;
            *reinterpret_cast<f32x4*>(pool + o) = best;
#pragma unroll
            for (int j = 0; j < 4; ++j) mxp = fmaxf(mxp, best[j]);
            if (code != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) code[o + j] = cd[j];
            }
        }
    }
    if (amax_out != nullptr) cvk_amax_publish_wg(mxo, amax_out, snapo);
    if (amax_pool != nullptr) { __syncthreads(); cvk_amax_publish_wg(mxp, amax_pool, snapp); }
}

// ---------------------------------------------------------------------------------------------- backward
// Shared walker: block b owns pixel rows [b*rows, (b+1)*rows); thread t owns channel vector (t % cvn) and walks
// pixels t/cvn, t/cvn + 256/cvn, ...  MODE 0: partial sums of g and g*xhat.  MODE 1: write dy, partial sums of dy.
template <int V, int MODE>
__global__ __launch_bounds__(256) void k_bn_bwd(const float* __restrict__ dout, PixMap dm, const float* __restrict__ y, int ldy,
                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                               const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                               float* __restrict__ dy, int ld_dy, float* __restrict__ part, int M, int C,
                                               int rows, int PB, int cchunk, int use_batch_stats, unsigned* __restrict__ amax) {
    typedef typename Vec<V>::T VT;
    __shared__ float red[2][256 * V];
    float mx = 0.f;                       // MODE 1 with amax: the largest |dy| this thread wrote
    const unsigned snap = MODE == 1 ? cvk_amax_snapshot(amax) : 0u;
    const int c0 = blockIdx.y * cchunk;
    const int cw = min(cchunk, C - c0);  // channels handled by this block column
    const int cvn = cw / V;               // vectors per pixel in this chunk (cw % V == 0 by construction)
    const int ppp = 256 / cvn;            // pixels per pass
    const int t = threadIdx.x;
    const bool active = t < cvn * ppp;
    const int cv = t % cvn, pr = t / cvn;
    const int c = c0 + cv * V;
    const int mbeg = blockIdx.x * rows, mend = min(M, mbeg + rows);

    float s0[V], s1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (active) {
        const VT sc = *reinterpret_cast<const VT*>(scale + c);
        const VT sh = *reinterpret_cast<const VT*>(shift + c);
        const VT mu = *reinterpret_cast<const VT*>(mean + c);
        const VT rs = *reinterpret_cast<const VT*>(rstd + c);
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
        for (int m = mbeg + pr; m < mend; m += ppp) {     // four iterations' loads in flight (the sums stay in order)
            const VT d = *reinterpret_cast<const VT*>(dout + dm.off(m) + c);
            const VT yy = *reinterpret_cast<const VT*>(y + (size_t)m * ldy + c);
            VT o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float yv = vget<V>(yy, j);
                const float g = (yv * vget<V>(sc, j) + vget<V>(sh, j) > 0.f) ? vget<V>(d, j) : 0.f;
                const float xh = (yv - vget<V>(mu, j)) * vget<V>(rs, j);
                if (MODE == 0) {
                    s0[j] += g;
                    s1[j] += g * xh;
                } else {
                    const float r = vget<V>(sc, j) * (g - k1[j] - xh * k2[j]);
                    vset<V>(o, j, r);
                    s0[j] += r;
                    mx = fmaxf(mx, fabsf(r));
                }
            }
            if (MODE == 1) *reinterpret_cast<VT*>(dy + (size_t)m * ld_dy + c) = o;
        }
    }
    if (MODE == 1 && amax != nullptr) cvk_amax_publish(mx, amax, snap);      // the split-operand consumer scales by the exact maximum (csrc/split_fmt.h)
    if (part == nullptr) return;
    // block combine: threads with equal cv, fixed order over pr
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
            part[(size_t)blockIdx.x * C + c0 + t * V + j] = a;
            if (MODE == 0) part[(size_t)(PB + blockIdx.x) * C + c0 + t * V + j] = b;
        }
    }
}

// MODE-1 pass that also writes the F(4,3) weight-grad's transformed output-gradient planes (csrc/wino4.hip):
//   E1 = dy0+dy1+dy2+dy3, E2 = dy0-dy1+dy2-dy3, E3 = dy0+2dy1+4dy2+8dy3, E4 = dy0-2dy1+4dy2-8dy3   per group of four columns,
// float E[4][N*H*ceil(W/4)][ld_dy], so that the weight-grad does not have to re-read dy for them.  One thread owns a
// 4-channel vector and walks column groups (tiles); a block covers `tiles` consecutive groups; partial column sums of dy
// per block as in MODE 1.
__global__ __launch_bounds__(256) void k_bn_bwd_dx_e(const float* __restrict__ dout, PixMap dm, const float* __restrict__ y, int ldy,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                    float* __restrict__ dy, int ld_dy, float* __restrict__ E,
                                                    float* __restrict__ part, int M, int C, int W, int Wt, int Mt, int tiles,
                                                    int cchunk, int use_batch_stats, int six_H, int six_Wtp, long six_rows,
                                                    unsigned* __restrict__ amax) {
    __shared__ float red[256 * 4];
    float mx = 0.f;                       // with amax: the largest |dy| this thread wrote
    const unsigned snap = cvk_amax_snapshot(amax);
    const int c0 = blockIdx.y * cchunk;
    const int cw = min(cchunk, C - c0);
    const int cvn = cw / 4;
    const int ppp = 256 / cvn;
    const int t = threadIdx.x;
    const bool active = t < cvn * ppp;
    const int cv = t % cvn, pr = t / cvn;
    const int c = c0 + cv * 4;
    const int tbeg = blockIdx.x * tiles, tend = min(Mt, tbeg + tiles);
    float s0[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + c);
        f32x4 k1 = {0.f, 0.f, 0.f, 0.f}, k2 = {0.f, 0.f, 0.f, 0.f};
        if (use_batch_stats) {
            const float invM = 1.f / (float)M;
#pragma unroll
            for (int j = 0; j < 4; ++j) { k1[j] = dbeta[c + j] * invM; k2[j] = dgamma[c + j] * invM; }
        }
        const size_t plane = (size_t)Mt * ld_dy;
        for (int tl = tbeg + pr; tl < tend; tl += ppp) {
            const int row = tl / Wt, xt = tl - row * Wt;
            const int m0 = row * W + 4 * xt;
            f32x4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (4 * xt + i < W) {
                    const int m = m0 + i;
                    const f32x4 d = *reinterpret_cast<const f32x4*>(dout + dm.off(m) + c);
                    const f32x4 yy = *reinterpret_cast<const f32x4*>(y + (size_t)m * ldy + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = (yy[j] * sc[j] + sh[j] > 0.f) ? d[j] : 0.f;
                        const float xh = (yy[j] - mu[j]) * rs[j];
                        const float v = sc[j] * (g - k1[j] - xh * k2[j]);
                        r[i][j] = v;
                        s0[j] += v;
                        mx = fmaxf(mx, fabsf(v));
                    }
                    *reinterpret_cast<f32x4*>(dy + (size_t)m * ld_dy + c) = r[i];
                }
            }
            const f32x4 a = r[0] + r[2], b = r[1] + r[3], cc = r[0] + 4.f * r[2], dd = 2.f * r[1] + 8.f * r[3];
            if (six_H == 0) {               // E1..E4 [4][Mt][ld_dy] for k_wgrad_wino4 (E0 / E5 are columns of dy itself)
                float* o = E + (size_t)tl * ld_dy + c;
                *reinterpret_cast<f32x4*>(o) = a + b;
                *reinterpret_cast<f32x4*>(o + plane) = a - b;
                *reinterpret_cast<f32x4*>(o + 2 * plane) = cc + dd;
                *reinterpret_cast<f32x4*>(o + 3 * plane) = cc - dd;
            } else {                        // E0..E5 in the padded plane layout of csrc/wgradp.hip (rows of C floats)
                const int n = row / six_H, yy = row - n * six_H;
                const size_t prow = (size_t)six_Wtp + ((size_t)n * (six_H + 2) + yy + 1) * six_Wtp + xt;
                const size_t ps = (size_t)six_rows * C;
                float* o = E + prow * C + c;
                if (six_rows < 0) {         // four planes E1..E4 only (round 6): E0 / E5 are columns of dy itself, the plane GEMM reads them from dy
                    const size_t p4 = (size_t)(-six_rows) * C;
                    *reinterpret_cast<f32x4*>(o) = a + b;
                    *reinterpret_cast<f32x4*>(o + p4) = a - b;
                    *reinterpret_cast<f32x4*>(o + 2 * p4) = cc + dd;
                    *reinterpret_cast<f32x4*>(o + 3 * p4) = cc - dd;
                } else {
                    *reinterpret_cast<f32x4*>(o) = r[0];
                    *reinterpret_cast<f32x4*>(o + ps) = a + b;
                    *reinterpret_cast<f32x4*>(o + 2 * ps) = a - b;
                    *reinterpret_cast<f32x4*>(o + 3 * ps) = cc + dd;
                    *reinterpret_cast<f32x4*>(o + 4 * ps) = cc - dd;
                    *reinterpret_cast<f32x4*>(o + 5 * ps) = r[3];
                }
            }
        }
    }
    if (amax != nullptr) cvk_amax_publish(mx, amax, snap);
    if (part == nullptr) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) red[t * 4 + j] = s0[j];
    __syncthreads();
    if (t < cvn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
            for (int p = 0; p < ppp; ++p) a += red[(p * cvn + t) * 4 + j];
            part[(size_t)blockIdx.x * C + c0 + t * 4 + j] = a;
        }
    }
}

__global__ __launch_bounds__(1024) void k_colsum_finalize(const float* __restrict__ part, int PB, int C, float* out0,
                                                         float* out1) {
    __shared__ double red[16][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    const int which = blockIdx.y;
    const float* src = part + (size_t)which * PB * C;
    double a = 0.0;
    if (c < C) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;       // four independent chains: the loads overlap (fixed order, deterministic)
        int p = g;
        for (; p + 48 < PB; p += 64) {
            a += (double)src[(size_t)p * C + c];
            a1 += (double)src[(size_t)(p + 16) * C + c];
            a2 += (double)src[(size_t)(p + 32) * C + c];
            a3 += (double)src[(size_t)(p + 48) * C + c];
        }
        for (; p < PB; p += 16) a += (double)src[(size_t)p * C + c];
        a = (a + a1) + (a2 + a3);
    }
    red[g][l] = a;
    __syncthreads();
    if (g == 0 && c < C) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][l];
        (which == 0 ? out0 : out1)[c] = (float)s;
    }
}

// several column-sum finalisations in one launch (the conv-bias gradients of a backward pass are not needed before its end: 23
// launches of ~8 us -> 1); jobs by value, blockIdx.y = job
struct ColsumJobsDev { cvk_colsum_job j[CVK_COLSUM_BATCH_MAX]; };
__global__ __launch_bounds__(1024) void k_colsum_finalize_batch(const ColsumJobsDev jobs) {
    __shared__ double red[16][64];
    const cvk_colsum_job& J = jobs.j[blockIdx.y];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    if (blockIdx.x * 64 >= J.C) return;                 // uniform per workgroup
    const float* src = J.part;
    const int PB = J.PB, C = J.C;
    double a = 0.0;
    if (c < C) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int p = g;
        for (; p + 48 < PB; p += 64) {
            a += (double)src[(size_t)p * C + c];
            a1 += (double)src[(size_t)(p + 16) * C + c];
            a2 += (double)src[(size_t)(p + 32) * C + c];
            a3 += (double)src[(size_t)(p + 48) * C + c];
        }
        for (; p < PB; p += 16) a += (double)src[(size_t)p * C + c];
        a = (a + a1) + (a2 + a3);
    }
    red[g][l] = a;
    __syncthreads();
    if (g == 0 && c < C) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][l];
        J.out[c] = (float)s;
    }
}

int bwd_rows(int M) {
    const int pb = cvk_bn_bwd_blocks(M);
    return cvk_cdiv(M, pb);
}

bool vec_ok(const void* a, const void* b, const void* c, int ld0, int ld1, int C, const PixMap* pm) {
    if (C % 4 || ld0 % 4 || ld1 % 4) return false;
    if (!cvk_aligned16(a) || !cvk_aligned16(b) || (c && !cvk_aligned16(c))) return false;
    if (pm && ((pm->sN | pm->sY | pm->sX) & 3)) return false;
    return true;
}

}  // namespace

extern "C" size_t cvk_bn_finalize_workspace_bytes(int P, int C) {
    if (P <= 0 || C <= 0) return 0;
    const int G = P < 32 ? 1 : (P / 32 < 64 ? P / 32 : 64);
    return (size_t)G * C * 3 * sizeof(double);
}

extern "C" int cvk_bn_finalize(const float* stats, int P, int M, int C, const float* gamma, const float* beta, float* mean,
                               float* rstd, float* scale, float* shift, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float momentum, float eps, void* workspace,
                               size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(stats && gamma && beta && mean && rstd && scale && shift && workspace, "cvk_bn_finalize: null pointer");
    CVK_CHECK_ARG(P == cvk_cdiv(M, CVK_STAT_ROWS) && C > 0 && M > 0, "cvk_bn_finalize: P=%d inconsistent with M=%d", P, M);
    CVK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "cvk_bn_finalize: running_mean/var must come together");
    CVK_CHECK_ARG((((uintptr_t)workspace) & 7u) == 0, "cvk_bn_finalize: workspace must be 8-byte aligned");
    if (workspace_bytes < cvk_bn_finalize_workspace_bytes(P, C)) {
        cvk_set_error("cvk_bn_finalize: workspace too small");
        return CVK_EWORKSPACE;
    }
    const int G = P < 32 ? 1 : (P / 32 < 64 ? P / 32 : 64);
    const int rows_per_g = cvk_cdiv(P, G);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_stats_l1, dim3(cvk_cdiv(C, 64), G), dim3(256), 0, s, stats, (double*)workspace, P, M, C, rows_per_g);
    hipLaunchKernelGGL(k_bn_stats_l2, dim3(cvk_cdiv(C, 64)), dim3(256), 0, s, (const double*)workspace, G, M, C, gamma, beta, mean,
                       rstd, scale, shift, running_mean, running_var, num_batches_tracked, momentum, eps);
    CVK_LAUNCH_RETURN("cvk_bn_finalize");
}

extern "C" int cvk_bn_finalize_counts(const float* stats, const float* counts, int P, int M, int C, const float* gamma, const float* beta,
                                      float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, float momentum, float eps, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    CVK_CHECK_ARG(stats && counts && gamma && beta && mean && rstd && scale && shift && workspace, "cvk_bn_finalize_counts: null pointer");
    CVK_CHECK_ARG(P > 0 && C > 0 && M > 0, "cvk_bn_finalize_counts: bad sizes");
    CVK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "cvk_bn_finalize_counts: running_mean/var must come together");
    CVK_CHECK_ARG((((uintptr_t)workspace) & 7u) == 0, "cvk_bn_finalize_counts: workspace must be 8-byte aligned");
    if (workspace_bytes < cvk_bn_finalize_workspace_bytes(P, C)) {
        cvk_set_error("cvk_bn_finalize_counts: workspace too small");
        return CVK_EWORKSPACE;
    }
    const int G = P < 32 ? 1 : (P / 32 < 64 ? P / 32 : 64);
    const int rows_per_g = cvk_cdiv(P, G);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_stats_l1_counts, dim3(cvk_cdiv(C, 64), G), dim3(256), 0, s, stats, counts, (double*)workspace, P, C, rows_per_g);
    hipLaunchKernelGGL(k_bn_stats_l2, dim3(cvk_cdiv(C, 64)), dim3(256), 0, s, (const double*)workspace, G, M, C, gamma, beta, mean,
                       rstd, scale, shift, running_mean, running_var, num_batches_tracked, momentum, eps);
    CVK_LAUNCH_RETURN("cvk_bn_finalize_counts");
}

extern "C" int cvk_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float* mean, float* rstd, float* scale, float* shift, int C, float eps, void* stream) {
    CVK_CHECK_ARG(gamma && beta && running_mean && running_var && mean && rstd && scale && shift && C > 0, "cvk_bn_eval_params: bad arguments");
    hipLaunchKernelGGL(k_bn_eval_params, dim3(cvk_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, mean, rstd, scale, shift, C, eps);
    CVK_LAUNCH_RETURN("cvk_bn_eval_params");
}

static int bn_relu_apply_launch(const char* who, const float* y, int ldy, const float* scale, const float* shift, cvk_view out, int N, int H,
                                int W, int C, void* amax, void* stream);
extern "C" int cvk_bn_relu_apply(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, int N, int H,
                                 int W, int C, void* stream) {
    return bn_relu_apply_launch("cvk_bn_relu_apply", y, ldy, scale, shift, out, N, H, W, C, nullptr, stream);
}
// ... that also combines the largest activation it writes into *amax_block (atomicMax of fp32 bit patterns; the caller zeroes the word): what
// cvk_absmax_f32 of the written view would return, without the extra pass (the opt-in fp16 split-operand transforms scale by it)
extern "C" int cvk_bn_relu_apply_amax(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, int N, int H,
                                      int W, int C, void* amax_block, void* stream) {
    CVK_CHECK_ARG(amax_block, "cvk_bn_relu_apply_amax: null amax_block");
    return bn_relu_apply_launch("cvk_bn_relu_apply_amax", y, ldy, scale, shift, out, N, H, W, C, amax_block, stream);
}
static int bn_relu_apply_launch(const char* who, const float* y, int ldy, const float* scale, const float* shift, cvk_view out, int N, int H,
                                int W, int C, void* amax, void* stream) {
    CVK_CHECK_ARG(y && scale && shift && out.ptr, "%s: null pointer", who);
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && (long)N * H * W < (1L << 31), "%s: bad shape", who);
    const int M = N * H * W;
    const PixMap om = make_map(out, H, W);
    const bool v4 = vec_ok(y, out.ptr, scale, ldy, 0, C, &om) && cvk_aligned16(shift);
    const long total = (long)M * (v4 ? C / 4 : C);
    const int cap = amax ? 2048 : 16384;        // with the maximum: one full round of longer-lived workgroups (the publish is per workgroup)
    const int blocks = (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
    if (v4)
        hipLaunchKernelGGL(k_bn_relu_apply<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, ldy, scale, shift, out.ptr, om, M, C, (unsigned*)amax);
    else
        hipLaunchKernelGGL(k_bn_relu_apply<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, ldy, scale, shift, out.ptr, om, M, C, (unsigned*)amax);
    CVK_LAUNCH_RETURN(who);
}

// BN-apply + ReLU + MaxPool2d(2,2) of the result (pool [N][H/2][W/2][C] dense, code optional: as cvk_maxpool2x2_fwd).  4-channel
// vector layout only (CVK_EINVAL otherwise: the caller then runs cvk_bn_relu_apply + cvk_maxpool2x2_fwd).
static int bn_relu_apply_pool_launch(const char* who, const float* y, int ldy, const float* scale, const float* shift, cvk_view out, float* pool,
                                     unsigned char* code, int N, int H, int W, int C, void* amax_out, void* amax_pool, void* stream);
extern "C" int cvk_bn_relu_apply_pool(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, float* pool,
                                      unsigned char* code, int N, int H, int W, int C, void* stream) {
    return bn_relu_apply_pool_launch("cvk_bn_relu_apply_pool", y, ldy, scale, shift, out, pool, code, N, H, W, C, nullptr, nullptr, stream);
}
// ... with the largest activation written through `out` and the largest pooled value combined into two words (either may be NULL), as
// cvk_bn_relu_apply_amax
extern "C" int cvk_bn_relu_apply_pool_amax(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, float* pool,
                                           unsigned char* code, int N, int H, int W, int C, void* amax_out, void* amax_pool, void* stream) {
    return bn_relu_apply_pool_launch("cvk_bn_relu_apply_pool_amax", y, ldy, scale, shift, out, pool, code, N, H, W, C, amax_out, amax_pool, stream);
}
static int bn_relu_apply_pool_launch(const char* who, const float* y, int ldy, const float* scale, const float* shift, cvk_view out, float* pool,
                                     unsigned char* code, int N, int H, int W, int C, void* amax_out, void* amax_pool, void* stream) {
    CVK_CHECK_ARG(y && scale && shift && out.ptr && pool, "%s: null pointer", who);
    CVK_CHECK_ARG(N > 0 && H >= 2 && W >= 2 && C > 0 && ldy >= C && (long)N * H * W < (1L << 31), "%s: bad shape", who);
    const PixMap om = make_map(out, H, W);
    const bool v4 = vec_ok(y, out.ptr, scale, ldy, 0, C, &om) && cvk_aligned16(shift) && cvk_aligned16(pool);
    CVK_CHECK_ARG(v4, "%s: needs the 4-channel vector layout", who);
    const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    const int cap = (amax_out || amax_pool) ? 2048 : 16384;
    const int blocks = (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
    hipLaunchKernelGGL(k_bn_relu_apply_pool, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, ldy, scale, shift, out.ptr, om, pool, code, N, H, W, C,
                       (unsigned*)amax_out, (unsigned*)amax_pool);
    CVK_LAUNCH_RETURN(who);
}

extern "C" int cvk_bn_bwd_blocks(int M) {
    if (M <= 0) return 0;
    int pb = cvk_cdiv(M, 16);
    pb = pb < 512 ? pb : 512;       // 2 blocks per CU keep the stream bandwidth-bound (4 per CU measured no faster); fewer partial rows keep the fp64 finalize short
    const int rows = cvk_cdiv(M, pb);
    return cvk_cdiv(M, rows);       // exactly the number of row blocks the kernels launch: every partial row gets written
}

static int bn_bwd_launch(int mode, cvk_view dout, const float* y, int ldy, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy,
                         float* part, int N, int H, int W, int C, int use_batch_stats, void* stream, const char* name, void* amax = nullptr) {
    CVK_CHECK_ARG(dout.ptr && y && scale && shift && mean && rstd, "%s: null pointer", name);
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && (long)N * H * W < (1L << 31), "%s: bad shape", name);
    const int M = N * H * W;
    const PixMap dm = make_map(dout, H, W);
    bool v4 = vec_ok(y, dout.ptr, scale, ldy, mode == 1 ? ld_dy : 0, C, &dm) && cvk_aligned16(shift) && cvk_aligned16(mean) && cvk_aligned16(rstd);
    if (mode == 1) v4 = v4 && cvk_aligned16(dy);
    const int PB = cvk_bn_bwd_blocks(M), rows = bwd_rows(M);
    const int nb = cvk_cdiv(M, rows);  // blocks actually needed (<= PB); unused partial rows are zero-filled below
    const int cchunk = v4 ? 1024 : 256;
    dim3 grid(nb, cvk_cdiv(C, cchunk));
    hipStream_t s = (hipStream_t)stream;
#define CVK_BWD(V_, MODE_)                                                                                              \
    hipLaunchKernelGGL((k_bn_bwd<V_, MODE_>), grid, dim3(256), 0, s, dout.ptr, dm, y, ldy, scale, shift, mean, rstd, dgamma, \
                       dbeta, dy, ld_dy, part, M, C, rows, PB, cchunk, use_batch_stats, (unsigned*)amax)
    if (mode == 0) { if (v4) CVK_BWD(4, 0); else CVK_BWD(1, 0); }
    else { if (v4) CVK_BWD(4, 1); else CVK_BWD(1, 1); }
#undef CVK_BWD
    CVK_LAUNCH_RETURN(name);
}

extern "C" int cvk_bn_bwd_reduce(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift,
                                 const float* mean, const float* rstd, float* part, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(part, "cvk_bn_bwd_reduce: null partial buffer");
    return bn_bwd_launch(0, dout, y, ldy, scale, shift, mean, rstd, nullptr, nullptr, nullptr, 0, part, N, H, W, C, 1, stream,
                         "cvk_bn_bwd_reduce");
}

extern "C" int cvk_bn_bwd_dx(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                             const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* dbias_part,
                             int N, int H, int W, int C, int use_batch_stats, void* stream) {
    CVK_CHECK_ARG(dy && ld_dy >= C, "cvk_bn_bwd_dx: bad dy");
    CVK_CHECK_ARG(!use_batch_stats || (dgamma && dbeta), "cvk_bn_bwd_dx: dgamma/dbeta required in training mode");
    return bn_bwd_launch(1, dout, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, dbias_part, N, H, W, C,
                         use_batch_stats, stream, "cvk_bn_bwd_dx");
}

// cvk_bn_bwd_dx that also combines the largest |dy| it writes into *amax_block (atomicMax of fp32 bit patterns; the caller zeroes the word):
// what cvk_absmax_f32(dy) would return, without the extra pass (the opt-in fp16 split-operand GEMMs scale by it: csrc/split_fmt.h)
extern "C" int cvk_bn_bwd_dx_amax(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                                  const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* dbias_part,
                                  int N, int H, int W, int C, int use_batch_stats, void* amax_block, void* stream) {
    CVK_CHECK_ARG(dy && ld_dy >= C && amax_block, "cvk_bn_bwd_dx_amax: bad dy / amax_block");
    CVK_CHECK_ARG(!use_batch_stats || (dgamma && dbeta), "cvk_bn_bwd_dx_amax: dgamma/dbeta required in training mode");
    return bn_bwd_launch(1, dout, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, dbias_part, N, H, W, C,
                         use_batch_stats, stream, "cvk_bn_bwd_dx_amax", amax_block);
}

// Fused variant of cvk_bn_bwd_dx for layers whose weight-grad runs through the transposed F(4,3): also writes the planes
// E1..E4 (float[4][N*H*ceil(W/4)][ld_dy], columns [C, ld_dy) zero when dy's are).  Needs the vectorised layout (C, ldy,
// ld_dy multiples of 4, 16-byte aligned pointers/strides); returns CVK_EINVAL otherwise — the caller then uses the plain
// pass and lets cvk_conv3x3_wgrad_wino4 transform dy itself.  `part` gets cvk_bn_bwd_e_blocks(N,H,W) partial rows.
extern "C" int cvk_bn_bwd_e_blocks(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int Mt = N * H * ((W + 3) / 4);
    const int pb = cvk_bn_bwd_blocks(N * H * W);
    const int tiles = cvk_cdiv(Mt, pb);
    return cvk_cdiv(Mt, tiles);
}

static int bn_bwd_dx_e_launch(const char* who, int six, cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                              const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E,
                              float* part, int N, int H, int W, int C, int use_batch_stats, void* amax, void* stream);
extern "C" int cvk_bn_bwd_dx_e(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                               const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E,
                               float* part, int N, int H, int W, int C, int use_batch_stats, void* stream) {
    return bn_bwd_dx_e_launch("cvk_bn_bwd_dx_e", 0, dout, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, E, part, N, H, W, C, use_batch_stats,
                              nullptr, stream);
}
// cvk_bn_bwd_dx_e / cvk_bn_bwd_dx_e6 (six != 0) that also leave the largest |dy| in an amax block (as cvk_bn_bwd_dx_amax)
extern "C" int cvk_bn_bwd_dx_e_amax(int six, cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                                    const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E,
                                    float* part, int N, int H, int W, int C, int use_batch_stats, void* amax_block, void* stream) {
    CVK_CHECK_ARG(amax_block, "cvk_bn_bwd_dx_e_amax: null amax block");
    return bn_bwd_dx_e_launch("cvk_bn_bwd_dx_e_amax", six == 2 ? 2 : (six ? 1 : 0), dout, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, E, part, N, H, W, C,
                              use_batch_stats, amax_block, stream);
}
static int bn_bwd_dx_e_launch(const char* who, int six, cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                              const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E,
                              float* part, int N, int H, int W, int C, int use_batch_stats, void* amax, void* stream) {
    if (six) {
        CVK_CHECK_ARG(dout.ptr && y && scale && shift && mean && rstd && dy && E && dgamma && dbeta, "%s: null pointer", who);
        CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && ld_dy == C && (long)N * H * W < (1L << 31), "%s: bad shape (ld_dy must equal C)", who);
        const PixMap dm6 = make_map(dout, H, W);
        const bool v46 = vec_ok(y, dout.ptr, scale, ldy, ld_dy, C, &dm6) && cvk_aligned16(shift) && cvk_aligned16(mean) &&
                         cvk_aligned16(rstd) && cvk_aligned16(dy) && cvk_aligned16(E);
        CVK_CHECK_ARG(v46, "%s: needs the 4-channel vector layout (use cvk_bn_bwd_dx)", who);
        const int M6 = N * H * W, Wt6 = (W + 3) / 4, Mt6 = N * H * Wt6;
        const int Wtp = (Wt6 + 7) / 8 * 8;
        const long rows = (long)N * (H + 2) * Wtp + 2L * Wtp;
        const int pb6 = cvk_bn_bwd_blocks(M6);
        const int tiles6 = cvk_cdiv(Mt6, pb6), nb6 = cvk_cdiv(Mt6, tiles6);
        dim3 grid6(nb6, cvk_cdiv(C, 1024));
        hipLaunchKernelGGL(k_bn_bwd_dx_e, grid6, dim3(256), 0, (hipStream_t)stream, dout.ptr, dm6, y, ldy, scale, shift, mean, rstd,
                           dgamma, dbeta, dy, ld_dy, E, part, M6, C, W, Wt6, Mt6, tiles6, 1024, use_batch_stats, H, Wtp, six == 2 ? -rows : rows, (unsigned*)amax);
        CVK_LAUNCH_RETURN(who);
    }
    CVK_CHECK_ARG(dout.ptr && y && scale && shift && mean && rstd && dy && E && dgamma && dbeta, "%s: null pointer", who);
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && ld_dy >= C && (long)N * H * W < (1L << 31), "%s: bad shape", who);
    const PixMap dm = make_map(dout, H, W);
    const bool v4 = vec_ok(y, dout.ptr, scale, ldy, ld_dy, C, &dm) && cvk_aligned16(shift) && cvk_aligned16(mean) &&
                    cvk_aligned16(rstd) && cvk_aligned16(dy) && cvk_aligned16(E);
    CVK_CHECK_ARG(v4, "%s: needs the 4-channel vector layout (use cvk_bn_bwd_dx)", who);
    const int M = N * H * W, Wt = (W + 3) / 4, Mt = N * H * Wt;
    const int pb = cvk_bn_bwd_blocks(M);
    const int tiles = cvk_cdiv(Mt, pb), nb = cvk_cdiv(Mt, tiles);
    const int cchunk = 1024;
    dim3 grid(nb, cvk_cdiv(C, cchunk));
    hipLaunchKernelGGL(k_bn_bwd_dx_e, grid, dim3(256), 0, (hipStream_t)stream, dout.ptr, dm, y, ldy, scale, shift, mean, rstd,
                       dgamma, dbeta, dy, ld_dy, E, part, M, C, W, Wt, Mt, tiles, cchunk, use_batch_stats, 0, 0, 0L, (unsigned*)amax);
    CVK_LAUNCH_RETURN(who);
}

// The same pass writing the SIX planes E0..E5 = A dy in the padded plane layout of csrc/wgradp.hip (cvk_wgradp_plane_rows rows of
// C floats per plane; the pad rows are zeroed by cvk_wgradp_zero_pads, not here).  ld_dy == C required.
extern "C" int cvk_bn_bwd_dx_e6(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                                const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E6,
                                float* part, int N, int H, int W, int C, int use_batch_stats, void* stream) {
    return bn_bwd_dx_e_launch("cvk_bn_bwd_dx_e6", 1, dout, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, E6, part, N, H, W, C, use_batch_stats,
                              nullptr, stream);
}

// ... writing only the FOUR planes E1..E4 (float[4][cvk_wgradp_plane_rows][C], padded layout, pad rows zeroed by cvk_wgradp_zero_pads4): E0 and E5
// are columns 4 xt and 4 xt + 3 of dy itself and cvk_wgradp_gemm_sm_dy reads them from dy — 1.0x instead of 1.5x the tensor written beside dy
extern "C" int cvk_bn_bwd_dx_e4p(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                                 const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E4p,
                                 float* part, int N, int H, int W, int C, int use_batch_stats, void* stream) {
    return bn_bwd_dx_e_launch("cvk_bn_bwd_dx_e4p", 2, dout, y, ldy, scale, shift, mean, rstd, dgamma, dbeta, dy, ld_dy, E4p, part, N, H, W, C, use_batch_stats,
                              nullptr, stream);
}

extern "C" int cvk_colsum_finalize_batch(const cvk_colsum_job* jobs, int n, void* stream) {
    CVK_CHECK_ARG(jobs && n > 0 && n <= CVK_COLSUM_BATCH_MAX, "cvk_colsum_finalize_batch: 1 <= n <= %d jobs", CVK_COLSUM_BATCH_MAX);
    ColsumJobsDev d;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        CVK_CHECK_ARG(jobs[i].part && jobs[i].out && jobs[i].PB > 0 && jobs[i].C > 0, "cvk_colsum_finalize_batch: bad job %d", i);
        d.j[i] = jobs[i];
        if (jobs[i].C > cmax) cmax = jobs[i].C;
    }
    hipLaunchKernelGGL(k_colsum_finalize_batch, dim3(cvk_cdiv(cmax, 64), n), dim3(1024), 0, (hipStream_t)stream, d);
    CVK_LAUNCH_RETURN("cvk_colsum_finalize_batch");
}

extern "C" int cvk_colsum_finalize(const float* part, int PB, int C, float* out0, float* out1, void* stream) {
    CVK_CHECK_ARG(part && out0 && PB > 0 && C > 0, "cvk_colsum_finalize: bad arguments");
    hipLaunchKernelGGL(k_colsum_finalize, dim3(cvk_cdiv(C, 64), out1 ? 2 : 1), dim3(1024), 0, (hipStream_t)stream, part, PB, C, out0, out1);
    CVK_LAUNCH_RETURN("cvk_colsum_finalize");
}
