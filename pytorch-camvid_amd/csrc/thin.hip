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
extern "C" int cvk_thin_fwd_supported(int Cin_ld, int Cout, int ldy) {
    return (Cin_ld == 64 && Cout <= 16 && ldy % 4 == 0 && ldy <= 16) ? 1 : 0;
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
    CVK_CHECK_ARG((long)H * W * 64 * 4 < (1L << 31), "cvk_conv3x3_thin_fwd: one image exceeds the 2 GiB buffer-addressing limit");
    const int XG = cvk_cdiv(W, 16);
    const int R = thin_rows_per_task(N, H, XG, thin_slots()), RC = cvk_cdiv(H, R), P = N * XG * RC;
    hipStream_t s = (hipStream_t)stream;
    if (stats)
        hipLaunchKernelGGL(k_thin_co_fwd<true>, dim3(cvk_cdiv(P, 4)), dim3(256), 0, s, x, w, bias, y, stats, counts, H, W, Cout, ldy, R, RC, XG, P);
    else
        hipLaunchKernelGGL(k_thin_co_fwd<false>, dim3(cvk_cdiv(P, 4)), dim3(256), 0, s, x, w, bias, y, stats, counts, H, W, Cout, ldy, R, RC, XG, P);
    CVK_LAUNCH_RETURN("cvk_conv3x3_thin_fwd");
}

extern "C" int cvk_thin_wgrad_supported(int Cin, int Cin_ld, int Cout, int ld_dy) {
    return (Cin == 64 && Cin_ld == 64 && Cout <= 16 && ld_dy <= 16) ? 1 : 0;
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
    return (size_t)cvk_cdiv(P, 4) * 36 * 256 * sizeof(float);
}

extern "C" int cvk_conv3x3_thin_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                                      int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dy && dw && workspace, "cvk_conv3x3_thin_wgrad: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ld_dy >= Cout, "cvk_conv3x3_thin_wgrad: bad shape");
    CVK_CHECK_ARG(cvk_thin_wgrad_supported(Cin, Cin_ld, Cout, ld_dy), "cvk_conv3x3_thin_wgrad: Cin=%d Cin_ld=%d Cout=%d ld_dy=%d is not a thin layer", Cin, Cin_ld, Cout, ld_dy);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(workspace), "cvk_conv3x3_thin_wgrad: x and the workspace must be 16-byte aligned");
    CVK_CHECK_ARG((long)H * W * 64 * 4 < (1L << 31), "cvk_conv3x3_thin_wgrad: one image exceeds the 2 GiB buffer-addressing limit");
    if (workspace_bytes < cvk_conv3x3_thin_wgrad_workspace_bytes(N, H, W, Cin_ld, Cout)) {
        cvk_set_error("cvk_conv3x3_thin_wgrad: workspace too small");
        return CVK_EWORKSPACE;
    }
    int R, RC, XG, P;
    thin_wgrad_plan(N, H, W, &R, &RC, &XG, &P);
    const int S = cvk_cdiv(P, 4);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_thin_co_wgrad, dim3(S), dim3(256), 0, s, x, dy, (float*)workspace, H, W, Cout, ld_dy, R, RC, XG, P);
    hipLaunchKernelGGL(k_thin_co_reduce, dim3(cvk_cdiv(Cout * 9 * 64, 64)), dim3(1024), 0, s, (const float*)workspace, dw, S, Cout);
    CVK_LAUNCH_RETURN("cvk_conv3x3_thin_wgrad");
}
