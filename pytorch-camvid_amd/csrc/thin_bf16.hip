// thin_bf16.hip — the THIN 3x3 convolutions of the bf16-storage mode (BASELINE.json configs[3]): the stem 3 -> 64 (reference
// models/unet.py:103 `BasicConv2d(in_channels, 64)`), the classifier head 64 -> class_num = 12 (models/unet.py:127), forward, and the
// head's data-grad 12 -> 64 (backward of train.py:131).  Rounds 2-4 ran them on k_conv_bf16s_strip, a kernel tiled for 32-channel K
// slices and 32-wide output-channel blocks: 3 (12) real channels in a 32-channel operand, 8-byte result stores, 0.52 ms for the three
// launches against ~0.2 ms of HBM time.  Here — the register-only design of thin.hip (fp32) on the bf16 matrix pipe:
//   * a wave owns 16 consecutive pixels of an image row and walks DOWN the image (input-row stationary: input row r feeds output rows
//     r+1, r, r-1 as kernel row 0, 1, 2; three output rows stay open); no activation goes through LDS, waves never synchronise;
//   * operands come straight from global memory in MFMA layout: the matrix instruction sums over k, so which channel a (lane group,
//     register) pair carries is free as long as both operands agree — lane group q = lane >> 4 loads the 16 (8) bytes [16q, 16q+16) of a
//     pixel's channel run with ONE load and that IS its k-chunk of the B operand; loads of the next input row are issued before the
//     MFMAs of the current one; out-of-frame taps are range-checked buffer loads with the offset forced out of range (no branches);
//   * the filter is packed once per step in lane order (cvk_pack_weight_thin_bf16) and shared by a workgroup's waves through LDS (the only
//     LDS use: one ds_read per MFMA; in registers it cost 72 VGPRs and two of the four resident waves per SIMD);
//   * head forward (64 real input channels): v_mfma_f32_16x16x32_bf16, 2 per tap; stem forward: v_mfma_f32_16x16x16_bf16 whose k = 16 is
//     the three taps of a kernel row x 4 padded channels (lane group q IS the column shift dx: one 8-byte load per lane and input row);
//     head data-grad: the same instruction, k = the 16 padded channels of dy, one per tap;
//   * 64-channel outputs: the rows of two neighbouring 16-channel blocks are interleaved (block b row m <-> channel 32 (b >> 1) +
//     8 (m >> 2) + 4 (b & 1) + (m & 3), applied by the weight pack), so a lane ends with EIGHT consecutive channels: 16-byte stores;
//   * BatchNorm statistics (models/unet.py:12) from the fp32 accumulators: per-lane sums down the column, one 16-lane reduction per
//     wave, one partial [sum | M2 about the partial mean | count] per wave -> cvk_bn_finalize_counts.
#include <type_traits>
#include <utility>
#include "cvk_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int N, class F, int... I>
__device__ __forceinline__ void tb_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void tb_static_for(F&& f) {
    tb_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ unsigned oob_if_not(bool ok, unsigned off) { return off | ((unsigned)(!ok) << 31); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t img_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ s16x4 ld8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
__device__ __forceinline__ float row16_sum(float v) {          // sum over the 16 lanes of a row, in every lane (fixed tree)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Column shifts without re-loading: the B operand of tap dx = 0 (2) is the pixel to the left (right) — the SAME registers one lane down
// (up) within the 16-lane row of pixels.  v_mov_dpp row_shr:1 / row_shl:1 with bound_ctrl off leaves the lane that has no source (lane 0 /
// lane 15 of the row) at `old`: the halo pixel from the neighbouring wave's column, which one extra (mostly masked) load brings.  A row of
// 16 pixels is loaded ONCE instead of three times (the first version issued three shifted loads per input row: 3x the L1 / TA traffic).
template <int CTRL> __device__ __forceinline__ unsigned dpp_or_old(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ bf16x8 shift_px(bf16x8 c, bf16x8 halo, bool left) {
    const u32x4 cu = __builtin_bit_cast(u32x4, c), hu = __builtin_bit_cast(u32x4, halo);
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = left ? dpp_or_old<0x111>(hu[i], cu[i]) : dpp_or_old<0x101>(hu[i], cu[i]);
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ s16x4 shift_px(s16x4 c, s16x4 halo, bool left) {
    const u32x2 cu = __builtin_bit_cast(u32x2, c), hu = __builtin_bit_cast(u32x2, halo);
    u32x2 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) r[i] = left ? dpp_or_old<0x111>(hu[i], cu[i]) : dpp_or_old<0x101>(hu[i], cu[i]);
    return __builtin_bit_cast(s16x4, r);
}

struct TbTask { int img, y0, y1, xg; };
__device__ __forceinline__ TbTask tb_task(int task, int XG, int RC, int R, int H) {
    TbTask t;
    const int ir = task / XG;
    t.xg = task - ir * XG;
    t.img = ir / RC;
    const int rc = ir - t.img * RC;
    t.y0 = rc * R;
    t.y1 = min(H, t.y0 + R);
    return t;
}

// channel of row m of 16-row block b under the interleave that gives a lane eight consecutive channels (see the file header)
__host__ __device__ __forceinline__ int tb_chan(int b, int m) { return 32 * (b >> 1) + 8 * (m >> 2) + 4 * (b & 1) + (m & 3); }

// ============================================================================================ head forward: 64 -> Cout <= 16
template <bool STATS>
__global__ __launch_bounds__(256, 3) void k_thinb_head_fwd(const __bf16* __restrict__ X, const bf16x8* __restrict__ Wp,
                                                       const float* __restrict__ bias, __bf16* __restrict__ Y, float* __restrict__ stats,
                                                       float* __restrict__ counts, int H, int W, int Cout, int ldy, int R, int RC, int XG, int P) {
    // the filter (18 A operands per lane = 72 registers) is shared by the workgroup's four waves through LDS in lane order: one
    // conflict-free ds_read_b128 per MFMA, and the waves stay under 128 registers (four per SIMD: the kernel lives on loads in flight)
    __shared__ bf16x8 wf[18 * 64];                     // [(dy*3 + dx)*2 + k half][lane]: W[co = l15][dy][dx][32 half + 8 q4 .. + 7]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 18 * 64; e += 256) wf[e] = Wp[e];
    __syncthreads();
    const int task = blockIdx.x * 4 + wave;
    if (task >= P) return;
    const int l15 = lane & 15, q4 = lane >> 4;
    const TbTask t = tb_task(task, XG, RC, R, H);
    const int px = t.xg * 16 + l15;
    const __amdgpu_buffer_rsrc_t xr = img_rsrc(X + (size_t)t.img * H * W * 64, (size_t)H * W * 64 * 2);
    const __amdgpu_buffer_rsrc_t yr = img_rsrc(Y + (size_t)t.img * H * W * ldy, (size_t)H * W * ldy * 2);
    // ring of three input rows, loads issued TWO rows ahead: [slot][0..1] = the lane's own pixel (k halves), [slot][2..3] = the halo pixel
    // of the row's end lanes (lane 0: px - 1, lane 15: px + 1; the other lanes request nothing)
    bf16x8 in[3][4];
    const int hx = l15 == 0 ? px - 1 : px + 1;
    const bool hlane = l15 == 0 || l15 == 15;
    auto load_row = [&](int yy, bf16x8 (&dst)[4]) {
        const bool rok = (unsigned)yy < (unsigned)H && yy <= t.y1;
        const unsigned off = oob_if_not(rok && (unsigned)px < (unsigned)W, (unsigned)((yy * W + px) * 64 + 8 * q4) * 2u);
        const unsigned hoff = oob_if_not(rok && hlane && (unsigned)hx < (unsigned)W, (unsigned)((yy * W + hx) * 64 + 8 * q4) * 2u);
        dst[0] = ld16(xr, off);
        dst[1] = ld16(xr, off + 64u);
        dst[2] = ld16(xr, hoff);
        dst[3] = ld16(xr, hoff + 64u);
    };
    f32x4v bs = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * q4 + i < Cout) bs[i] = bias[4 * q4 + i];
    }
    f32x4v s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const bool pok = px < W;
    const bool sok = pok && 4 * q4 < Cout;
    const unsigned obase = (unsigned)(px * ldy + 4 * q4) * 2u;
    f32x4v acc[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) acc[o] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int steps = t.y1 - t.y0 + 2;                 // step `it` handles input row y0 - 1 + it
    load_row(t.y0 - 1, in[0]);
    load_row(t.y0, in[1]);
    for (int ib = 0; ib < steps; ib += 6) {
        tb_static_for<6>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int it = ib + k;
            if (it < steps) {
                load_row(t.y0 + it + 1, in[(k + 2) % 3]);
                bf16x8 b[3][2];                                         // [dx][k half]
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    b[1][kh] = in[k % 3][kh];
                    b[0][kh] = shift_px(in[k % 3][kh], in[k % 3][2 + kh], true);
                    b[2][kh] = shift_px(in[k % 3][kh], in[k % 3][2 + kh], false);
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yo = t.y0 + it - dy;                      // output row fed through kernel row dy
                    if (yo >= t.y0 && yo < t.y1) {
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int kh = 0; kh < 2; ++kh)
                                acc[(k + 3 - dy) % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[((dy * 3 + d) * 2 + kh) * 64 + lane], b[d][kh],
                                                                                               acc[(k + 3 - dy) % 3], 0, 0, 0);
                    }
                }
                const int yd = t.y0 + it - 2;                           // complete after its kernel row 2
                if (yd >= t.y0) {
                    constexpr int sd = (k + 1) % 3;
                    const f32x4v v = acc[sd];
                    acc[sd] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    const f32x4v o = v + bs;
                    const bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), yr, oob_if_not(sok, obase + (unsigned)(yd * W) * ldy * 2u), 0, 0);
                    if (STATS) {
                        const f32x4v z = pok ? v : f32x4v{0.f, 0.f, 0.f, 0.f};
                        s1 += z;
                        s2 += z * z;
                    }
                }
            }
        });
    }
    if (STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1[i] = row16_sum(s1[i]); s2[i] = row16_sum(s2[i]); }
        if (l15 == 0) {
            const float cnt = (float)((t.y1 - t.y0) * min(16, W - t.xg * 16));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * q4 + i;
                if (c < Cout) {
                    const float m2 = s2[i] - s1[i] * s1[i] / cnt;       // sums exclude the bias (shift invariance)
                    stats[(size_t)task * Cout + c] = s1[i] + cnt * bs[i];
                    stats[(size_t)(P + task) * Cout + c] = m2 > 0.f ? m2 : 0.f;
                }
            }
            if (q4 == 0) counts[task] = cnt;
        }
    }
}

// ================================================================ 64 output channels on v_mfma_f32_16x16x16_bf16 (stem forward, head data-grad)
// NG = k groups per kernel row: 1 (stem: k = 3 column shifts x 4 padded input channels, lane group q4 is the shift dx; q4 = 3 is zero) or
// 3 (head data-grad: one group per shift dx, k = the 16 padded channels of dy, lane group q4 = channels 4 q4 .. 4 q4 + 3).
// A operands: Wp [(dy * NG + g) * 4 + b][lane] (4 bf16), block b row m <-> output channel tb_chan(b, m).
template <int NG, bool STATS>
__global__ __launch_bounds__(256, (STATS || NG == 3) ? 3 : 4) void k_thinb_wide(const __bf16* __restrict__ X, const s16x4* __restrict__ Wp, const float* __restrict__ bias,
                                                   __bf16* __restrict__ Y, float* __restrict__ stats, float* __restrict__ counts, int H, int W,
                                                   int ldx, int ldy, int R, int RC, int XG, int P) {
    __shared__ s16x4 wf[3 * NG * 4 * 64];              // the filter in lane order, shared by the four waves (see k_thinb_head_fwd)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 3 * NG * 4 * 64; e += 256) wf[e] = Wp[e];
    __syncthreads();
    const int task = blockIdx.x * 4 + wave;
    if (task >= P) return;
    const int l15 = lane & 15, q4 = lane >> 4;
    const TbTask t = tb_task(task, XG, RC, R, H);
    const int px = t.xg * 16 + l15;
    const __amdgpu_buffer_rsrc_t xr = img_rsrc(X + (size_t)t.img * H * W * ldx, (size_t)H * W * ldx * 2);
    const __amdgpu_buffer_rsrc_t yr = img_rsrc(Y + (size_t)t.img * H * W * ldy, (size_t)H * W * ldy * 2);
    // ring of three input rows, loads issued two rows ahead.  Stem (NG = 1): [slot][0] = pixel px + q4 - 1, channels 0..3 (the column
    // shift is the lane group).  Data-grad (NG = 3): [slot][0] = the lane's own pixel, channels 4 q4 .. 4 q4 + 3, [slot][1] = the halo pixel
    // of the row's end lanes; the shifted operands come from the neighbouring lanes (shift_px).
    s16x4 in[3][2];
    const int hx = l15 == 0 ? px - 1 : px + 1;
    const bool hlane = l15 == 0 || l15 == 15;
    auto load_row = [&](int yy, s16x4 (&dst)[2]) {
        const bool rok = (unsigned)yy < (unsigned)H && yy <= t.y1;
        if (NG == 1) {
            const int xx = px + q4 - 1;
            dst[0] = ld8(xr, oob_if_not(rok && (unsigned)xx < (unsigned)W && q4 < 3, (unsigned)((yy * W + xx) * ldx) * 2u));
        } else {
            dst[0] = ld8(xr, oob_if_not(rok && (unsigned)px < (unsigned)W, (unsigned)((yy * W + px) * ldx + 4 * q4) * 2u));
            dst[1] = ld8(xr, oob_if_not(rok && hlane && (unsigned)hx < (unsigned)W, (unsigned)((yy * W + hx) * ldx + 4 * q4) * 2u));
        }
    };
    // this lane's 16 output channels: block pair p (blocks 2p, 2p + 1) -> channels 32 p + 8 q4 + 0 .. 7
    float bs[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[p][e] = bias != nullptr ? bias[32 * p + 8 * q4 + e] : 0.f;
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[p][e] = 0.f; s2[p][e] = 0.f; }
    const bool pok = px < W;
    const unsigned obase = (unsigned)(px * ldy + 8 * q4) * 2u;
    f32x4v acc[3][4];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[o][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int steps = t.y1 - t.y0 + 2;
    load_row(t.y0 - 1, in[0]);
    load_row(t.y0, in[1]);
    for (int ib = 0; ib < steps; ib += 6) {
        tb_static_for<6>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int it = ib + k;
            if (it < steps) {
                load_row(t.y0 + it + 1, in[(k + 2) % 3]);
                s16x4 bop[NG];
                if (NG == 1) bop[0] = in[k % 3][0];
                else {
                    bop[1 % NG] = in[k % 3][0];
                    bop[0] = shift_px(in[k % 3][0], in[k % 3][1], true);
                    bop[2 % NG] = shift_px(in[k % 3][0], in[k % 3][1], false);
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yo = t.y0 + it - dy;
                    if (yo >= t.y0 && yo < t.y1) {
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                acc[(k + 3 - dy) % 3][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf[((dy * NG + g) * 4 + b) * 64 + lane], bop[g],
                                                                                                  acc[(k + 3 - dy) % 3][b], 0, 0, 0);
                    }
                }
                const int yd = t.y0 + it - 2;
                if (yd >= t.y0) {
                    constexpr int sd = (k + 1) % 3;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = acc[sd][2 * p + (e >> 2)][e & 3];
                        acc[sd][2 * p] = f32x4v{0.f, 0.f, 0.f, 0.f};
                        acc[sd][2 * p + 1] = f32x4v{0.f, 0.f, 0.f, 0.f};
                        bf16x8 ob;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ob[e] = (__bf16)(v[e] + bs[p][e]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ob), yr,
                                                               oob_if_not(pok, obase + (unsigned)(yd * W) * ldy * 2u + 64u * p), 0, 0);
                        if (STATS) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float z = pok ? v[e] : 0.f;
                                s1[p][e] += z;
                                s2[p][e] += z * z;
                            }
                        }
                    }
                }
            }
        });
    }
    if (STATS) {
        const float cnt = (float)((t.y1 - t.y0) * min(16, W - t.xg * 16));
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = row16_sum(s1[p][e]), b = row16_sum(s2[p][e]);
                if (l15 == 0) {
                    const int c = 32 * p + 8 * q4 + e;
                    const float m2 = b - a * a / cnt;
                    stats[(size_t)task * 64 + c] = a + cnt * bs[p][e];
                    stats[(size_t)(P + task) * 64 + c] = m2 > 0.f ? m2 : 0.f;
                }
            }
        if (lane == 0) counts[task] = cnt;
    }
}

// ---- filter packs: fp32 master [Cout][3][3][Cin] -> the A operands in lane order ----------------------------------------------------
// mode 1 (head forward):   out[(dy*3+dx)*2+kh][lane][8] = W[co = lane & 15][dy][dx][32 kh + 8 (lane >> 4) + e]
// mode 2 (stem forward):   out[dy*4+b][lane][4]        = W[tb_chan(b, lane & 15)][dy][dx = lane >> 4][cin = e]            (dx < 3, e < Cin)
// mode 3 (head data-grad): out[((dy*3+dx)*4+b][lane][4] = W[co = 4 (lane >> 4) + e][2 - dy][2 - dx][ci = tb_chan(b, lane & 15)]   (co < Cout)
__global__ void k_pack_thin_bf16(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int mode, int total) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (mode == 1) {
            const int e = i & 7, lane = (i >> 3) & 63, f = i >> 9;
            const int kh = f & 1, tap = f >> 1, co = lane & 15, ci = 32 * kh + 8 * (lane >> 4) + e;
            if (co < Cout && ci < Cin) v = w[((size_t)co * 9 + tap) * Cin + ci];
        } else {
            const int e = i & 3, lane = (i >> 2) & 63, f = i >> 8;
            const int b = f & 3, m = lane & 15, q = lane >> 4;
            if (mode == 2) {
                const int dy = f >> 2, co = tb_chan(b, m);
                if (q < 3 && e < Cin && co < Cout) v = w[((size_t)co * 9 + dy * 3 + q) * Cin + e];
            } else {
                const int tap = f >> 2, co = 4 * q + e, ci = tb_chan(b, m);
                if (co < Cout && ci < Cin) v = w[((size_t)co * 9 + (8 - tap)) * Cin + ci];
            }
        }
        out[i] = (__bf16)v;
    }
}

int thinb_slots() {
    static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 0 ? n : 256; }();
    // three rounds of (up to) four resident waves per SIMD: with 1.5 rounds of long tasks the second round ran half empty (first version:
    // 6000 tasks of 29 rows = 1.46 rounds)
    return cus * 4 * 4 * 3;
}
int thinb_rows(int N, int H, int XG) {
    const long segs = (long)N * H * XG;
    int R = (int)((segs + thinb_slots() - 1) / thinb_slots());
    if (R < 8) R = 8;                      // two halo rows per task: at least 8 own rows
    if (R > H) R = H;
    return R;
}

}  // namespace

// mode of the thin bf16 kernels for a layer: 1 head forward (64 input channels in a dense 64-channel tensor, Cout <= 16, Cout % 4 == 0),
// 2 stem forward (Cin <= 4 real channels in a tensor of pitch ld_in >= 4, 64 output channels), 3 head data-grad (the forward layer had
// 64 input and Cout <= 16 output channels: dy has <= 16 real channels in pitch ld_in, dX is a dense 64-channel tensor); 0: not thin.
// (Cin, Cout) are those of the FORWARD layer in every case.
extern "C" int cvk_thin_bf16_mode(int Cin, int Cout, int ld_in, int ld_out, int dgrad) {
    if (!dgrad) {
        if (Cin == 64 && ld_in == 64 && Cout >= 4 && Cout <= 16 && Cout % 4 == 0 && ld_out % 4 == 0 && ld_out >= Cout) return 1;
        if (Cin >= 1 && Cin <= 4 && ld_in >= 4 && ld_in % 4 == 0 && Cout == 64 && ld_out % 8 == 0 && ld_out >= 64) return 2;
        return 0;
    }
    if (Cin == 64 && Cout >= 1 && Cout <= 16 && ld_in >= 16 && ld_in % 4 == 0 && ld_out % 8 == 0 && ld_out >= 64) return 3;
    return 0;
}

extern "C" int cvk_thin_bf16_stat_partials(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int XG = cvk_cdiv(W, 16);
    return N * XG * cvk_cdiv(H, thinb_rows(N, H, XG));
}

extern "C" size_t cvk_thin_bf16_pack_elems(int mode) { return mode == 1 ? 18 * 64 * 8 : (mode == 2 ? 12 * 64 * 4 : (mode == 3 ? 36 * 64 * 4 : 0)); }

// w: the forward layer's fp32 weights, physical [Cout][3][3][Cin]
extern "C" int cvk_pack_weight_thin_bf16(const float* w, void* out, int Cout, int Cin, int mode, void* stream) {
    CVK_CHECK_ARG(w && out && Cout > 0 && Cin > 0 && mode >= 1 && mode <= 3, "cvk_pack_weight_thin_bf16: bad arguments");
    CVK_CHECK_ARG(mode == 2 ? (Cin <= 4 && Cout == 64) : (Cin == 64 && Cout <= 16), "cvk_pack_weight_thin_bf16: Cin=%d Cout=%d does not fit mode %d", Cin, Cout, mode);
    const int total = (int)cvk_thin_bf16_pack_elems(mode);
    hipLaunchKernelGGL(k_pack_thin_bf16, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)out, Cout, Cin, mode, total);
    CVK_LAUNCH_RETURN("cvk_pack_weight_thin_bf16");
}

// y = conv3x3(x, pack) (+ bias) (+ BatchNorm statistics partials: stats [2][P][Cout_k], counts [P], P = cvk_thin_bf16_stat_partials, Cout_k
// = Cout for mode 1, 64 otherwise).  x bf16 [N,H,W,ld_in], y bf16 [N,H,W,ld_out].  Cout: the real output channels of mode 1 (ignored otherwise).
extern "C" int cvk_conv3x3_thin_bf16(const void* x, const void* wpack, const float* bias, void* y, float* stats, float* counts, int N, int H,
                                     int W, int ld_in, int Cout, int ld_out, int mode, void* stream) {
    CVK_CHECK_ARG(x && wpack && y && N > 0 && H > 0 && W > 0 && mode >= 1 && mode <= 3, "cvk_conv3x3_thin_bf16: bad arguments");
    CVK_CHECK_ARG((stats == nullptr) == (counts == nullptr), "cvk_conv3x3_thin_bf16: stats and counts come together");
    CVK_CHECK_ARG(mode != 3 || stats == nullptr, "cvk_conv3x3_thin_bf16: the data-grad has no statistics");
    CVK_CHECK_ARG(mode == 1 ? (ld_in == 64 && Cout >= 4 && Cout <= 16 && Cout % 4 == 0 && ld_out % 4 == 0 && ld_out >= Cout)
                            : (ld_in >= (mode == 2 ? 4 : 16) && ld_in % 4 == 0 && ld_out % 8 == 0 && ld_out >= 64),
                  "cvk_conv3x3_thin_bf16: shape does not fit mode %d", mode);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(wpack) && cvk_aligned16(y), "cvk_conv3x3_thin_bf16: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)H * W * (ld_in > ld_out ? ld_in : ld_out) * 2 < (1L << 31), "cvk_conv3x3_thin_bf16: one image exceeds 2 GiB");
    const int XG = cvk_cdiv(W, 16), R = thinb_rows(N, H, XG), RC = cvk_cdiv(H, R);
    const long Pl = (long)N * XG * RC;
    CVK_CHECK_ARG(Pl < (1L << 30), "cvk_conv3x3_thin_bf16: too many tasks");
    const int P = (int)Pl;
    const dim3 grid((unsigned)cvk_cdiv(P, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1) {
        if (stats) hipLaunchKernelGGL((k_thinb_head_fwd<true>), grid, block, 0, s, (const __bf16*)x, (const bf16x8*)wpack, bias, (__bf16*)y, stats, counts, H, W, Cout, ld_out, R, RC, XG, P);
        else hipLaunchKernelGGL((k_thinb_head_fwd<false>), grid, block, 0, s, (const __bf16*)x, (const bf16x8*)wpack, bias, (__bf16*)y, stats, counts, H, W, Cout, ld_out, R, RC, XG, P);
    } else if (mode == 2) {
        if (stats) hipLaunchKernelGGL((k_thinb_wide<1, true>), grid, block, 0, s, (const __bf16*)x, (const s16x4*)wpack, bias, (__bf16*)y, stats, counts, H, W, ld_in, ld_out, R, RC, XG, P);
        else hipLaunchKernelGGL((k_thinb_wide<1, false>), grid, block, 0, s, (const __bf16*)x, (const s16x4*)wpack, bias, (__bf16*)y, stats, counts, H, W, ld_in, ld_out, R, RC, XG, P);
    } else {
        hipLaunchKernelGGL((k_thinb_wide<3, false>), grid, block, 0, s, (const __bf16*)x, (const s16x4*)wpack, bias, (__bf16*)y, stats, counts, H, W, ld_in, ld_out, R, RC, XG, P);
    }
    CVK_LAUNCH_RETURN("cvk_conv3x3_thin_bf16");
}
