// thin.hip — the 3x3 convolutions with a THIN channel dimension: the stem (3 -> 64, reference models/unet.py:103
// `BasicConv2d(in_channels, 64)` and models/segnet.py first block) and the classifier head (64 -> class_num = 12,
// models/unet.py:127 / the last SegNet block), forward, data-grad and weight-grad (backward of train.py:131).
//
// At 360x480 these five launches took 1.5 ms of a 38 ms step in kernels tiled for wide layers (a 32- or 64-wide channel tile for
// 12 channels, a 32-channel K slice for 3): 0.15-0.3 of the fp32 matrix peak and 5x their HBM time.  Here the thin dimension
// is one 16-row side of v_mfma_f32_16x16x4_f32 (or its k = 4), nothing goes through LDS except the head's filter, and every
// wave works alone:
//   * a wave owns 16 consecutive pixels of an image row (forward kernels) or 4 (weight-grad: the pixels are the MFMA's k) and
//     walks DOWN the image: the three input rows of a 3x3 window are a ring of registers, a row is loaded once per wave, its
//     loads are issued two rows ahead of their first use;
//   * operands come straight from global memory in MFMA layout.  The matrix instruction sums over k, so WHICH channel a
//     (lane group, register) pair carries is free as long as both operands agree: lane group q = lane >> 4 loads the 16 bytes
//     [4q, 4q+4) of a 64-byte channel run with ONE dwordx4 and register r of it is used as "k = q" of MFMA r.  A 16-pixel x
//     64-channel operand is four 1 KiB loads (contiguous 64-byte runs), no transpose, no shuffle;
//   * output orientation D[channel][pixel]: a lane ends up with four consecutive channels of one pixel -> 16-byte stores;
//   * BatchNorm statistics of the forward kernels (models/unet.py:12) from the accumulators: per-lane sums down the column,
//     one 16-lane reduction per wave, one partial [sum | M2 about the partial mean | count] per wave (cvk_bn_finalize_counts).
// Out-of-frame taps (zero padding, ragged right edge, rows outside the image) are range-checked buffer loads with the offset
// forced out of range: they return 0, no branches.
#include "conv_tile.h"

#include <type_traits>
#include <utility>

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int N, class F, int... I>
__device__ __forceinline__ void t_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void t_static_for(F&& f) {
    t_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* base, size_t image_floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(image_floats * 4), 0x00020000);
}
__device__ __forceinline__ float buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// task -> (image, row chunk, column group); consecutive tasks are horizontal neighbours (shared halo columns in L1/L2)
struct ThinTask {
    int img, y0, y1, xg;
};
__device__ __forceinline__ ThinTask thin_task(int task, int XG, int RC, int R, int H) {
    ThinTask t;
    const int ir = task / XG;
    t.xg = task - ir * XG;
    t.img = ir / RC;
    const int rc = ir - t.img * RC;
    t.y0 = rc * R;
    t.y1 = min(H, t.y0 + R);
    return t;
}

// ============================================================================================ head forward: 64 -> Cout <= 16
// D[co][px] += Wt[co][tap][k] * X[px + tap][k]: 9 taps x 64 channels = 144 MFMAs per 16 pixels, four accumulator chains.
// The filter (144 A-operand values per lane) is shared by the workgroup's waves through LDS in lane-linear order: one
// conflict-free ds_read_b128 per four MFMAs.

template <bool STATS>
__global__ __launch_bounds__(256, 2) void k_thin_co_fwd(const float* __restrict__ X, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ Y,
                                                        float* __restrict__ stats, float* __restrict__ counts, int H, int W,
                                                        int Cout, int ldy, int R, int RC, int XG, int P) {
    __shared__ f32x4 wl[36 * 64];                       // [tap][j][lane]: Wt[co = lane & 15][tap][16 j + 4 (lane >> 4) + 0..3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    for (int e = tid; e < 36 * 64; e += 256) {
        const int l = e & 63, tj = e >> 6;              // tj = tap * 4 + j
        const int co = l & 15, lq = l >> 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (co < Cout) v = *reinterpret_cast<const f32x4*>(Wt + ((size_t)co * 9 + (tj >> 2)) * 64 + 16 * (tj & 3) + 4 * lq);
        wl[e] = v;
    }
    __syncthreads();
    const int task = blockIdx.x * 4 + wave;
    if (task >= P) return;
    const ThinTask t = thin_task(task, XG, RC, R, H);
    const int px = t.xg * 16 + n;
    const __amdgpu_buffer_rsrc_t xr = image_rsrc(X + (size_t)t.img * H * W * 64, (size_t)H * W * 64);
    const __amdgpu_buffer_rsrc_t yr = image_rsrc(Y + (size_t)t.img * H * W * ldy, (size_t)H * W * ldy);

    // INPUT-row stationary: input row r (3 dx shifts x 64 channels = 48 registers, double buffered) feeds the three output
    // rows r+1, r, r-1 as kernel row 0, 1, 2; three output rows x four accumulator chains stay open.  (Output-row stationary
    // needs a ring of three input rows = 144 registers: spills at two waves per SIMD.)
    f32x4 in[2][3][4];                                  // [buffer][dx][j]: x[row][px + dx - 1][16 j + 4 q + 0..3]
    auto load_row = [&](int yy, f32x4 (&dst)[3][4]) {
        const bool rok = (unsigned)yy < (unsigned)H && yy <= t.y1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = px + d - 1;
            const unsigned off = oob_unless(rok && (unsigned)xx < (unsigned)W, (unsigned)((yy * W + xx) * 64 + 4 * q) * 4u);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[d][j] = buf_load16(xr, off + 64u * j);
        }
    };
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * q + i < Cout) bs[i] = bias[4 * q + i];
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const bool pok = px < W;
    const unsigned obase = (unsigned)(px * ldy + 4 * q) * 4u;
    const bool sok = pok && 4 * q < ldy;
    f32x4 acc[3][4];                                    // [output row slot = (row - y0) mod 3][chain j]
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // step `it` handles input row y0 - 1 + it, it = 0 .. rows + 1
    const int steps = t.y1 - t.y0 + 2;
    load_row(t.y0 - 1, in[0]);
    for (int ib = 0; ib < steps; ib += 6) {
        t_static_for<6>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int it = ib + k;
            if (it < steps) {
                load_row(t.y0 + it, in[(k + 1) & 1]);               // next input row: a whole step (144 MFMAs) ahead
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yo = t.y0 + it - dy;                  // output row fed through kernel row dy
                    constexpr int slot_base = k + 3;
                    if (yo >= t.y0 && yo < t.y1) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const f32x4 a = wl[((dy * 3 + d) * 4 + j) * 64 + lane];
                                const f32x4 b = in[k & 1][d][j];
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[(slot_base - dy) % 3][j] = mfma4(a[r], b[r], acc[(slot_base - dy) % 3][j]);
                            }
                        }
                    }
                }
                const int yd = t.y0 + it - 2;                       // complete after its kernel row 2
                if (yd >= t.y0) {
                    constexpr int sd = (k + 1) % 3;                 // (it - 2) mod 3
                    const f32x4 v = (acc[sd][0] + acc[sd][1]) + (acc[sd][2] + acc[sd][3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[sd][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 o = v + bs;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, oob_unless(sok, obase + (unsigned)(yd * W) * ldy * 4u), 0, 0);
                    if (STATS) {
                        const f32x4 z = pok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                        s1 += z;
                        s2 += z * z;
                    }
                }
            }
        });
    }
    if (STATS) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s1[i] += __shfl_xor(s1[i], m, 64);
                s2[i] += __shfl_xor(s2[i], m, 64);
            }
        }
        if (n == 0) {
            const float cnt = (float)((t.y1 - t.y0) * min(16, W - t.xg * 16));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * q + i;
                if (c < Cout) {
                    const float m2 = s2[i] - s1[i] * s1[i] / cnt;       // sums exclude the bias (shift invariance)
                    stats[(size_t)task * Cout + c] = s1[i] + cnt * bs[i];
                    stats[(size_t)(P + task) * Cout + c] = m2 > 0.f ? m2 : 0.f;
                }
            }
            if (q == 0) counts[task] = cnt;
        }
    }
}

// ====================================================================================== head weight-grad: dW[Cout <= 16][9][64]
// dW[co][tap][ci] = sum_p dY[p][co] * X[p + tap][ci]: the pixels are the MFMA's k.  A wave owns a 4-pixel-wide column strip:
// A = dY quad [co][4 px] (one dword per lane), B = the X quad shifted by the tap, one dwordx4 per lane = [4 px][64 ci] with
// register r <-> channels 4 n + r; 36 independent 16 x 16 accumulators (9 taps x 4 channel residues).  Ring of five input
// rows (loads issued two rows ahead).  The workgroup's four waves are summed through LDS in a fixed order; one partial per
// workgroup in lane-linear order, reduced (fixed order) and un-permuted by k_thin_reduce.
constexpr int TW_RING = 5;

__global__ __launch_bounds__(256, 2) void k_thin_co_wgrad(const float* __restrict__ X, const float* __restrict__ DY,
                                                          float* __restrict__ part, int H, int W, int Cout, int ld_dy, int R,
                                                          int RC, int XG, int P) {
    __shared__ f32x4 red[36 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int task = blockIdx.x * 4 + wave;
    f32x4 acc[9][4];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tp][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (task < P) {
        const ThinTask t = thin_task(task, XG, RC, R, H);
        const int px = t.xg * 4 + q;
        const __amdgpu_buffer_rsrc_t xr = image_rsrc(X + (size_t)t.img * H * W * 64, (size_t)H * W * 64);
        const __amdgpu_buffer_rsrc_t gr = image_rsrc(DY + (size_t)t.img * H * W * ld_dy, (size_t)H * W * ld_dy);
        f32x4 rows[TW_RING][3];                         // [ring slot][dx]: x[row][px + dx - 1][4 n + 0..3]
        float gq[TW_RING];                              // dY[row][px][n]
        auto load_row = [&](int yy, f32x4 (&dst)[3]) {
            const bool rok = (unsigned)yy < (unsigned)H && yy <= t.y1;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int xx = px + d - 1;
                dst[d] = buf_load16(xr, oob_unless(rok && (unsigned)xx < (unsigned)W, (unsigned)((yy * W + xx) * 64 + 4 * n) * 4u));
            }
        };
        auto load_g = [&](int yy) {
            return buf_load4(gr, oob_unless(yy < t.y1 && px < W && n < Cout, (unsigned)((yy * W + px) * ld_dy + n) * 4u));
        };
        // slot of input row r = (r - y0 + 1) mod 5; slot of dY row y = (y - y0) mod 5
        load_row(t.y0 - 1, rows[0]);
        load_row(t.y0, rows[1]);
        load_row(t.y0 + 1, rows[2]);
        load_row(t.y0 + 2, rows[3]);
        gq[0] = load_g(t.y0);
        gq[1] = load_g(t.y0 + 1);
        gq[2] = load_g(t.y0 + 2);
        for (int yb = t.y0; yb < t.y1; yb += TW_RING) {
            t_static_for<TW_RING>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const int y = yb + k;
                if (y < t.y1) {
                    load_row(y + 3, rows[(k + 4) % TW_RING]);
                    gq[(k + 3) % TW_RING] = load_g(y + 3);
                    const float a = gq[k];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const f32x4 b = rows[(k + dy) % TW_RING][d];
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[dy * 3 + d][r] = mfma4(a, b[r], acc[dy * 3 + d][r]);
                        }
                }
            });
        }
    }
    // waves 3, 2, 1 hand their sums down: (2 += 3 is skipped: fixed order 0 + (1 + (2 + 3)) would need two buffers) -> 0 + 1 + 2 + 3
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 36; ++a) red[a * 64 + lane] = acc[a >> 2][a & 3];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int a = 0; a < 36; ++a) acc[a >> 2][a & 3] += red[a * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        f32x4* dst = reinterpret_cast<f32x4*>(part) + (size_t)blockIdx.x * 36 * 64;
#pragma unroll
        for (int a = 0; a < 36; ++a) dst[a * 64 + lane] = acc[a >> 2][a & 3];
    }
}

// dW[co][tap][ci] = sum_s part[s][tap * 4 + (ci & 3)][lane = (co >> 2) * 16 + (ci >> 2)][co & 3]; 64 outputs x 16 groups of
// partials per workgroup, fixed order
__global__ __launch_bounds__(1024) void k_thin_co_reduce(const float* __restrict__ part, float* __restrict__ dw, int S, int Cout) {
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;                  // output index (co * 9 + tap) * 64 + ci
    const int total = Cout * 9 * 64;
    float s = 0.f;
    if (e < total) {
        const int ci = e & 63, ct = e >> 6, tap = ct % 9, co = ct / 9;
        const size_t src = ((size_t)(tap * 4 + (ci & 3)) * 64 + (co >> 2) * 16 + (ci >> 2)) * 4 + (co & 3);
        float s0 = 0.f, s1 = 0.f;
        int p = g;
        for (; p + 16 < S; p += 32) {
            s0 += part[(size_t)p * 36 * 256 + src];
            s1 += part[(size_t)(p + 16) * 36 * 256 + src];
        }
        if (p < S) s0 += part[(size_t)p * 36 * 256 + src];
        s = s0 + s1;
    }
    red[g][l] = s;
    __syncthreads();
    if (g == 0 && e < total) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[i][l];
        dw[e] = a;
    }
}

// ============================================================================ thin INPUT: Cin_ld = 4 NV <= 12 -> wide Cout
// The stem (3 -> 64, Cin_ld = 4) and the head's data-grad (12 -> 64 on the rotated/transposed pack, Cin_ld = 12).  A tap's
// channels are the k of ONE MFMA (NV = 1) or of NV of them: lane group q carries channels [q NV, q NV + NV).  D[co][px]: a
// workgroup column (blockIdx.y) owns 64 output channels = 4 row blocks, the filter (36 NV values per lane) lives in registers.
// Input-row stationary as k_thin_co_fwd; a row is only 3 NV registers, so four are kept in flight.
template <int NV, bool STATS>
__global__ __launch_bounds__(256, 2) void k_thin_ci_fwd(const float* __restrict__ X, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ Y,
                                                        float* __restrict__ stats, float* __restrict__ counts, int H, int W,
                                                        int Cout, int ldy, int R, int RC, int XG, int P) {
    constexpr int LD = 4 * NV;
    constexpr int RING = NV == 1 ? 4 : 2;               // input rows in registers: loads run RING - 1 steps (of 36 NV MFMAs) ahead
    constexpr int UN = RING == 4 ? 12 : 6;              // lcm(RING, 3 output-row slots): every index below is static
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int task = blockIdx.x * 4 + wave;
    if (task >= P) return;
    const int cb = blockIdx.y * 64;
    const ThinTask t = thin_task(task, XG, RC, R, H);
    const int px = t.xg * 16 + n;
    const __amdgpu_buffer_rsrc_t xr = image_rsrc(X + (size_t)t.img * H * W * LD, (size_t)H * W * LD);
    const __amdgpu_buffer_rsrc_t yr = image_rsrc(Y + (size_t)t.img * H * W * ldy, (size_t)H * W * ldy);

    float wa[4][9][NV];                                 // Wt[cb + 16 nb + n][tap][q NV + r]
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int co = cb + nb * 16 + n;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < NV; ++r) wa[nb][tp][r] = co < Cout ? Wt[((size_t)co * 9 + tp) * LD + q * NV + r] : 0.f;
    }
    float in[RING][3][NV];                              // [ring slot][dx][r]: x[row][px + dx - 1][q NV + r]
    auto load_row = [&](int yy, float (&dst)[3][NV]) {
        const bool rok = (unsigned)yy < (unsigned)H && yy <= t.y1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = px + d - 1;
            const unsigned off = oob_unless(rok && (unsigned)xx < (unsigned)W, (unsigned)((yy * W + xx) * LD + q * NV) * 4u);
#pragma unroll
            for (int r = 0; r < NV; ++r) dst[d][r] = buf_load4(xr, off + 4u * r);
        }
    };
    f32x4 bs[4], s1[4], s2[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        bs[nb] = s1[nb] = s2[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (cb + nb * 16 + 4 * q + i < Cout) bs[nb][i] = bias[cb + nb * 16 + 4 * q + i];
        }
    }
    const bool pok = px < W;
    const unsigned obase = (unsigned)(px * ldy + cb + 4 * q) * 4u;
    f32x4 acc[3][4];                                    // [output row slot = (row - y0) mod 3][row block]
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[o][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // step `it` handles input row y0 - 1 + it (ring slot it mod RING), it = 0 .. rows + 1
    const int steps = t.y1 - t.y0 + 2;
#pragma unroll
    for (int j = 0; j < RING - 1; ++j) load_row(t.y0 - 1 + j, in[j]);
    for (int ib = 0; ib < steps; ib += UN) {
        t_static_for<UN>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int it = ib + k;
            if (it < steps) {
                load_row(t.y0 + it + RING - 2, in[(k + RING - 1) % RING]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yo = t.y0 + it - dy;                  // output row fed through kernel row dy
                    constexpr int slot_base = k + 3;
                    if (yo >= t.y0 && yo < t.y1) {
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int r = 0; r < NV; ++r)
#pragma unroll
                                for (int nb = 0; nb < 4; ++nb)
                                    acc[(slot_base - dy) % 3][nb] = mfma4(wa[nb][dy * 3 + d][r], in[k % RING][d][r], acc[(slot_base - dy) % 3][nb]);
                    }
                }
                const int yd = t.y0 + it - 2;                       // complete after its kernel row 2
                if (yd >= t.y0) {
                    constexpr int sd = (k + 1) % 3;                 // (it - 2) mod 3
                    const unsigned orow = obase + (unsigned)(yd * W) * ldy * 4u;
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const f32x4 v = acc[sd][nb];
                        acc[sd][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const f32x4 o = v + bs[nb];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr,
                                                               oob_unless(pok && cb + nb * 16 + 4 * q < ldy, orow + 64u * nb), 0, 0);
                        if (STATS) {
                            const f32x4 z = pok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                            s1[nb] += z;
                            s2[nb] += z * z;
                        }
                    }
                }
            }
        });
    }
    if (STATS) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s1[nb][i] += __shfl_xor(s1[nb][i], m, 64);
                    s2[nb][i] += __shfl_xor(s2[nb][i], m, 64);
                }
        }
        if (n == 0) {
            const float cnt = (float)((t.y1 - t.y0) * min(16, W - t.xg * 16));
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = cb + nb * 16 + 4 * q + i;
                    if (c < Cout) {
                        const float m2 = s2[nb][i] - s1[nb][i] * s1[nb][i] / cnt;      // sums exclude the bias (shift invariance)
                        stats[(size_t)task * Cout + c] = s1[nb][i] + cnt * bs[nb][i];
                        stats[(size_t)(P + task) * Cout + c] = m2 > 0.f ? m2 : 0.f;
                    }
                }
            if (q == 0 && blockIdx.y == 0) counts[task] = cnt;
        }
    }
}

// ================================================================================= stem weight-grad: dW[Cout][9][Cin <= 4]
// The pixels are k again (4-pixel column strips).  B = the dY quad: one dwordx4 per lane = [4 px][64 co], register r <-> output
// channels 4 n + r.  A = the X quad for ONE kernel row, all three kernel columns at once: row 4 dx + c of the 16 <-> (kernel
// column dx, input channel c) — x[px - 1 + dx][c] for lane group k = pixel sits 16 B x (pixel + dx) + 4 B x c from the strip's
// left neighbour, i.e. 4 B x row: one dword per lane, rows 12..15 unused.  12 MFMAs per image row and 64 output channels; rows
// are processed in blocks of four, the next block's 6 + 16 registers in flight under the current one.
constexpr int TS_ROWS = 4;

__global__ __launch_bounds__(256, 4) void k_thin_ci_wgrad(const float* __restrict__ X, const float* __restrict__ DY,
                                                          float* __restrict__ part, int H, int W, int Cout, int ld_dy, int R,
                                                          int RC, int XG, int P) {
    __shared__ f32x4 red[12 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int task = blockIdx.x * 4 + wave;
    const int cb = blockIdx.y * 64;
    f32x4 acc[3][4];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[dy][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (task < P) {
        const ThinTask t = thin_task(task, XG, RC, R, H);
        const int px = t.xg * 4 + q;
        const __amdgpu_buffer_rsrc_t xr = image_rsrc(X + (size_t)t.img * H * W * 4, (size_t)H * W * 4);
        const __amdgpu_buffer_rsrc_t gr = image_rsrc(DY + (size_t)t.img * H * W * ld_dy, (size_t)H * W * ld_dy);
        const int xx = px - 1 + (n >> 2);
        const bool xok = n < 12 && (unsigned)xx < (unsigned)W;
        const bool gok = px < W && cb + 4 * n < Cout;
        float xa[2][TS_ROWS + 2];                       // x rows yb-1 .. yb+4 of the block
        f32x4 gb[2][TS_ROWS];                           // dY rows yb .. yb+3
        auto load_block = [&](int yb, float (&xd)[TS_ROWS + 2], f32x4 (&gd)[TS_ROWS]) {
#pragma unroll
            for (int j = 0; j < TS_ROWS + 2; ++j) {
                const int yy = yb - 1 + j;
                xd[j] = buf_load4(xr, oob_unless(xok && (unsigned)yy < (unsigned)H && yy <= t.y1, (unsigned)((yy * W + xx) * 4 + (n & 3)) * 4u));
            }
#pragma unroll
            for (int j = 0; j < TS_ROWS; ++j) {
                const int yy = yb + j;
                gd[j] = buf_load16(gr, oob_unless(gok && yy < t.y1, (unsigned)((yy * W + px) * ld_dy + cb + 4 * n) * 4u));
            }
        };
        load_block(t.y0, xa[0], gb[0]);
        for (int yb = t.y0; yb < t.y1; yb += 2 * TS_ROWS) {
            t_static_for<2>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const int y = yb + k * TS_ROWS;
                if (y < t.y1) {
                    load_block(y + TS_ROWS, xa[k ^ 1], gb[k ^ 1]);
#pragma unroll
                    for (int j = 0; j < TS_ROWS; ++j)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[dy][r] = mfma4(xa[k][j + dy], gb[k][j][r], acc[dy][r]);
                }
            });
        }
    }
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {                       // wave 0 adds waves 1, 2, 3 in that order
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 12; ++a) red[a * 64 + lane] = acc[a >> 2][a & 3];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int a = 0; a < 12; ++a) acc[a >> 2][a & 3] += red[a * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        f32x4* dst = reinterpret_cast<f32x4*>(part) + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 12 * 64;
#pragma unroll
        for (int a = 0; a < 12; ++a) dst[a * 64 + lane] = acc[a >> 2][a & 3];
    }
}

// dW[co][tap = 3 dy + dx][ci] = sum_s part[s][co >> 6][dy * 4 + (co & 3)][lane = dx * 16 + ((co & 63) >> 2)][ci]
__global__ __launch_bounds__(1024) void k_thin_ci_reduce(const float* __restrict__ part, float* __restrict__ dw, int S, int CB, int Cout,
                                                         int Cin) {
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;                  // output index (co * 9 + tap) * Cin + ci
    const int total = Cout * 9 * Cin;
    float s = 0.f;
    if (e < total) {
        const int ci = e % Cin, ct = e / Cin, tap = ct % 9, co = ct / 9;
        const int dy = tap / 3, dx = tap - 3 * dy;
        const size_t src = ((size_t)(co >> 6) * 12 + dy * 4 + (co & 3)) * 256 + (size_t)(dx * 16 + ((co & 63) >> 2)) * 4 + ci;
        const size_t stride = (size_t)CB * 12 * 256;
        float s0 = 0.f, s1 = 0.f;
        int p = g;
        for (; p + 16 < S; p += 32) {
            s0 += part[(size_t)p * stride + src];
            s1 += part[(size_t)(p + 16) * stride + src];
        }
        if (p < S) s0 += part[(size_t)p * stride + src];
        s = s0 + s1;
    }
    red[g][l] = s;
    __syncthreads();
    if (g == 0 && e < total) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[i][l];
        dw[e] = a;
    }
}

// rows per task: the fewest rounds of `slots` concurrent waves, then the fewest rows per wave (each task also loads two halo rows)
int thin_rows_per_task(int N, int H, int groups, int slots) {
    long best = -1;
    int bestR = H;
    for (int rc = 1; rc <= H && rc <= 64; ++rc) {
        const int R = cvk_cdiv(H, rc);
        if (R < 8 && rc > 1) break;
        const long tasks = (long)N * groups * cvk_cdiv(H, R);
        const long rounds = (tasks + slots - 1) / slots;
        const long cost = rounds * (R + 3);
        if (best < 0 || cost < best) {
            best = cost;
            bestR = R;
        }
    }
    return bestR;
}

int thin_slots() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus * 8;                                     // two waves per SIMD
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------- C ABI
static bool thin_co_shape(int Cin_ld, int Cout, int ldy) { return Cin_ld == 64 && Cout <= 16 && ldy % 4 == 0 && ldy <= 16; }
static bool thin_ci_shape(int Cin_ld, int Cout, int ldy) { return Cin_ld >= 4 && Cin_ld <= 12 && Cin_ld % 4 == 0 && Cout >= 32 && ldy % 4 == 0; }

extern "C" int cvk_thin_fwd_supported(int Cin_ld, int Cout, int ldy) {
    return (thin_co_shape(Cin_ld, Cout, ldy) || thin_ci_shape(Cin_ld, Cout, ldy)) ? 1 : 0;
}

extern "C" int cvk_thin_stat_partials(int N, int H, int W, int Cin_ld) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int XG = cvk_cdiv(W, 16);
    const int R = thin_rows_per_task(N, H, XG, thin_slots());
    return N * XG * cvk_cdiv(H, R);
}

extern "C" int cvk_conv3x3_thin_fwd(const float* x, const float* w, const float* bias, float* y, float* stats, float* counts, int N,
                                    int H, int W, int Cin_ld, int Cout, int ldy, void* stream) {
    CVK_CHECK_ARG(x && w && y, "cvk_conv3x3_thin_fwd: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout, "cvk_conv3x3_thin_fwd: bad shape");
    CVK_CHECK_ARG(cvk_thin_fwd_supported(Cin_ld, Cout, ldy), "cvk_conv3x3_thin_fwd: Cin_ld=%d Cout=%d ldy=%d is not a thin layer", Cin_ld, Cout, ldy);
    CVK_CHECK_ARG((stats == nullptr) == (counts == nullptr), "cvk_conv3x3_thin_fwd: stats and counts go together");
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(w) && cvk_aligned16(y), "cvk_conv3x3_thin_fwd: x, w and y must be 16-byte aligned");
    CVK_CHECK_ARG((long)H * W * (Cin_ld > ldy ? Cin_ld : ldy) * 4 < (1L << 31), "cvk_conv3x3_thin_fwd: one image exceeds the 2 GiB buffer-addressing limit");
    const int XG = cvk_cdiv(W, 16);
    const int R = thin_rows_per_task(N, H, XG, thin_slots()), RC = cvk_cdiv(H, R), P = N * XG * RC;
    hipStream_t s = (hipStream_t)stream;
    if (!thin_co_shape(Cin_ld, Cout, ldy)) {
        const dim3 grid(cvk_cdiv(P, 4), cvk_cdiv(Cout, 64));
        // (12 input channels with statistics would spill at two waves per SIMD; no layer of the reference needs it)
        CVK_CHECK_ARG(!(stats && Cin_ld > 8), "cvk_conv3x3_thin_fwd: statistics are fused for Cin_ld <= 8 only");
#define CVK_THIN_CI(NV_, ST_) hipLaunchKernelGGL((k_thin_ci_fwd<NV_, ST_>), grid, dim3(256), 0, s, x, w, bias, y, stats, counts, H, W, Cout, ldy, R, RC, XG, P)
        if (Cin_ld == 4 && stats) CVK_THIN_CI(1, true);
        else if (Cin_ld == 4) CVK_THIN_CI(1, false);
        else if (Cin_ld == 8 && stats) CVK_THIN_CI(2, true);
        else if (Cin_ld == 8) CVK_THIN_CI(2, false);
        else CVK_THIN_CI(3, false);
#undef CVK_THIN_CI
        CVK_LAUNCH_RETURN("cvk_conv3x3_thin_fwd");
    }
    if (stats)
        hipLaunchKernelGGL(k_thin_co_fwd<true>, dim3(cvk_cdiv(P, 4)), dim3(256), 0, s, x, w, bias, y, stats, counts, H, W, Cout, ldy, R, RC, XG, P);
    else
        hipLaunchKernelGGL(k_thin_co_fwd<false>, dim3(cvk_cdiv(P, 4)), dim3(256), 0, s, x, w, bias, y, stats, counts, H, W, Cout, ldy, R, RC, XG, P);
    CVK_LAUNCH_RETURN("cvk_conv3x3_thin_fwd");
}

static bool thin_co_wshape(int Cin, int Cin_ld, int Cout, int ld_dy) { return Cin == 64 && Cin_ld == 64 && Cout <= 16 && ld_dy <= 16; }
static bool thin_ci_wshape(int Cin, int Cin_ld, int Cout, int ld_dy) { return Cin <= 4 && Cin_ld == 4 && Cout >= 32 && ld_dy % 4 == 0; }

extern "C" int cvk_thin_wgrad_supported(int Cin, int Cin_ld, int Cout, int ld_dy) {
    return (thin_co_wshape(Cin, Cin_ld, Cout, ld_dy) || thin_ci_wshape(Cin, Cin_ld, Cout, ld_dy)) ? 1 : 0;
}

static void thin_wgrad_plan(int N, int H, int W, int* R, int* RC, int* XG, int* P) {
    *XG = cvk_cdiv(W, 4);
    *R = thin_rows_per_task(N, H, *XG, thin_slots());
    *RC = cvk_cdiv(H, *R);
    *P = N * *XG * *RC;
}

extern "C" size_t cvk_conv3x3_thin_wgrad_workspace_bytes(int N, int H, int W, int Cin_ld, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    int R, RC, XG, P;
    thin_wgrad_plan(N, H, W, &R, &RC, &XG, &P);
    if (Cin_ld == 4) return (size_t)cvk_cdiv(P, 4) * cvk_cdiv(Cout, 64) * 12 * 256 * sizeof(float);
    return (size_t)cvk_cdiv(P, 4) * 36 * 256 * sizeof(float);
}

extern "C" int cvk_conv3x3_thin_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                                      int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dy && dw && workspace, "cvk_conv3x3_thin_wgrad: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ld_dy >= Cout, "cvk_conv3x3_thin_wgrad: bad shape");
    CVK_CHECK_ARG(cvk_thin_wgrad_supported(Cin, Cin_ld, Cout, ld_dy), "cvk_conv3x3_thin_wgrad: Cin=%d Cin_ld=%d Cout=%d ld_dy=%d is not a thin layer", Cin, Cin_ld, Cout, ld_dy);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(workspace), "cvk_conv3x3_thin_wgrad: x and the workspace must be 16-byte aligned");
    CVK_CHECK_ARG((long)H * W * (Cin_ld > ld_dy ? Cin_ld : ld_dy) * 4 < (1L << 31), "cvk_conv3x3_thin_wgrad: one image exceeds the 2 GiB buffer-addressing limit");
    if (workspace_bytes < cvk_conv3x3_thin_wgrad_workspace_bytes(N, H, W, Cin_ld, Cout)) {
        cvk_set_error("cvk_conv3x3_thin_wgrad: workspace too small");
        return CVK_EWORKSPACE;
    }
    int R, RC, XG, P;
    thin_wgrad_plan(N, H, W, &R, &RC, &XG, &P);
    const int S = cvk_cdiv(P, 4);
    hipStream_t s = (hipStream_t)stream;
    if (Cin_ld == 4) {
        CVK_CHECK_ARG(cvk_aligned16(dy), "cvk_conv3x3_thin_wgrad: dy must be 16-byte aligned");
        const int CB = cvk_cdiv(Cout, 64);
        hipLaunchKernelGGL(k_thin_ci_wgrad, dim3(S, CB), dim3(256), 0, s, x, dy, (float*)workspace, H, W, Cout, ld_dy, R, RC, XG, P);
        hipLaunchKernelGGL(k_thin_ci_reduce, dim3(cvk_cdiv(Cout * 9 * Cin, 64)), dim3(1024), 0, s, (const float*)workspace, dw, S, CB, Cout, Cin);
        CVK_LAUNCH_RETURN("cvk_conv3x3_thin_wgrad");
    }
    hipLaunchKernelGGL(k_thin_co_wgrad, dim3(S), dim3(256), 0, s, x, dy, (float*)workspace, H, W, Cout, ld_dy, R, RC, XG, P);
    hipLaunchKernelGGL(k_thin_co_reduce, dim3(cvk_cdiv(Cout * 9 * 64, 64)), dim3(1024), 0, s, (const float*)workspace, dw, S, Cout);
    CVK_LAUNCH_RETURN("cvk_conv3x3_thin_wgrad");
}
