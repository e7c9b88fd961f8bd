// conv_bf16.hip — opt-in reduced-precision variant of the direct 3x3 convolution (forward and data-grad):
// operands are rounded to bf16 on the way to LDS and multiplied on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate); activations, weights, bias, BatchNorm statistics and everything in HBM
// stay fp32.  This is the "bf16 + MFMA im2col path" of BASELINE.json configs[3]; it is NOT used by the fp32 headline
// (set per network with pytorch_camvid_amd.set_conv_precision(net, "bf16")).
//
// Same implicit GEMM and pipeline as conv3x3.hip (two LDS stages, two register stages, range-checked buffer loads,
// one basic block per K step); the differences: LDS rows hold 32 bf16 (64 B, padded to 80 B so the 16 rows of a
// ds_read_b128 lane group hit 16 distinct 16-byte slots), one ds_read_b128 is one 8-element K fragment, and a K slice
// is 2 MFMAs per 32x32 tile instead of 16 — the kernel is staging-bound (L2 -> LDS), not MFMA-bound.
#include "conv_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int LDB = 40;   // LDS row pitch in bf16 elements (80 B)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(unsigned, v);
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool STATS>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k_conv3x3_igemm_bf16(
    const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ Y,
    float* __restrict__ stats, int M, int H, int W, int Cin, int Cout, int ldy, int Ktot, int P, int tilesN) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int RP = NT / 8;
    constexpr int NA = BM / RP, NB = BN / RP;
    constexpr int STAGE = (BM + BN) * LDB;   // bf16 elements
    static_assert(NA >= 1 && NB >= 1 && BM % RP == 0 && BN % RP == 0, "tile/threads mismatch");
    static_assert(!STATS || TM == 2, "BN statistics granule is 64 rows per wave");

    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    const int tile = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, M * Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, Cout * Ktot * 4, 0x00020000);

    const int kv = tid & 7, r0 = tid >> 3;
    unsigned aoff[NA], amask[NA], boff[NB];
    const int HW = H * W;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + r0 + i * RP;
        unsigned mask = 0;
        if (m < M) {
            const int n = m / HW, rem = m - n * HW;
            const int y = rem / W, x = rem - y * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) mask |= 1u << t;
            }
        }
        amask[i] = mask;
        aoff[i] = (unsigned)(m < M ? m : 0) * (unsigned)Cin * 4u + kv * 16u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = n0 + r0 + i * RP;
        boff[i] = co < Cout ? (unsigned)co * (unsigned)Ktot * 4u + kv * 16u : OOB;
    }

    f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    int lk = 0, ltap = 0, lcib = 0;     // next slice to load (uniform): K base, tap, channel base (Cin % 32 == 0)

    auto issue_loads = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
        const int t3 = (ltap * 11) >> 5;
        const int dy = t3 - 1, dx = ltap - 3 * t3 - 1;
        const unsigned sh = (unsigned)(((dy * W + dx) * Cin + lcib) * 4);
        const unsigned bit = 1u << ltap;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = buf_load16(xr, oob_unless((amask[i] & bit) != 0, aoff[i] + sh));
        const unsigned kb = lk < Ktot ? (unsigned)lk * 4u : OOB;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = buf_load16(wr, (boff[i] + kb) | ((boff[i] | kb) & OOB));
        lk += BK;
        lcib += BK;
        const int w1 = lcib >= Cin;
        lcib = w1 ? 0 : lcib;
        ltap += w1;
    };
    auto store_stage = [&](__bf16* dst, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            uint2 v = {pack_bf16(ra[i][0], ra[i][1]), pack_bf16(ra[i][2], ra[i][3])};
            *reinterpret_cast<uint2*>(&dst[(r0 + i * RP) * LDB + kv * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            uint2 v = {pack_bf16(rb[i][0], rb[i][1]), pack_bf16(rb[i][2], rb[i][3])};
            *reinterpret_cast<uint2*>(&dst[BM * LDB + (r0 + i * RP) * LDB + kv * 4]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // A/B fragment of v_mfma_f32_32x32x16_bf16: lane (r = l & 31, h = l >> 5) holds row r, k = 8h .. 8h+7
    auto mma_kk = [&](const __bf16* arow, const __bf16* brow, int kk) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const bf16x8*>(arow + t * 32 * LDB + kk * 16);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const bf16x8*>(brow + t * 32 * LDB + kk * 16);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    };

    const int nK = Ktot / BK;
    issue_loads(ra0, rb0);
    store_stage(smem, ra0, rb0);
    issue_loads(ra0, rb0);
    issue_loads(ra1, rb1);
    __syncthreads();
    const int aro = (wm * TM * 32 + li) * LDB + lh * 8;
    const int bro = BM * LDB + (wn * TN * 32 + li) * LDB + lh * 8;
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + STAGE;
#define CVK_BSTEP(cur, nxt, RA, RB)        \
    do {                                   \
        mma_kk(cur + aro, cur + bro, 0);   \
        store_stage(nxt, RA, RB);          \
        issue_loads(RA, RB);               \
        mma_kk(cur + aro, cur + bro, 1);   \
        __syncthreads();                   \
    } while (0)
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_BSTEP(buf0, buf1, ra0, rb0);
        CVK_BSTEP(buf1, buf0, ra1, rb1);
    }
    if (ks < nK) CVK_BSTEP(buf0, buf1, ra0, rb0);
#undef CVK_BSTEP

    // ---- epilogue (identical to the fp32 kernel): + bias, NHWC store, fused BatchNorm statistics partials
    const int rowbase = m0 + wm * TM * 32;
    const bool full = (m0 + BM <= M) && (n0 + BN <= ldy) && (n0 + BN <= Cout);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + li;
        const float bv = (bias != nullptr && col < Cout) ? bias[col] : 0.f;
        float s = 0.f;
        if (full) {
            float* yp = Y + (size_t)(rowbase + 4 * lh) * ldy + col;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[tm][tn][r] + bv;
                    acc[tm][tn][r] = v;
                    s += v;
                    yp[(size_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * ldy] = v;
                }
        } else {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rowbase + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v = acc[tm][tn][r] + bv;
                    acc[tm][tn][r] = v;
                    if (row < M) {
                        s += v;
                        if (col < ldy) Y[(size_t)row * ldy + col] = v;
                    }
                }
        }
        if (STATS) {
            const int cnt = min(64, M - rowbase);
            if (cnt > 0) {
                s += __shfl_xor(s, 32, 64);
                const float mean = s / (float)cnt;
                float q = 0.f;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rowbase + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float d = acc[tm][tn][r] - mean;
                        if (full || row < M) q += d * d;
                    }
                q += __shfl_xor(q, 32, 64);
                const int prow = rowbase / CVK_STAT_ROWS;
                if (col < Cout) {
                    if (lh == 0) stats[(size_t)prow * Cout + col] = s;
                    else stats[(size_t)(P + prow) * Cout + col] = q;
                }
            }
        }
    }
}

}  // namespace

extern "C" int cvk_conv3x3_fwd_bf16(const float* x, const float* w, const float* bias, float* y, float* stats, int N, int H,
                                    int W, int Cin, int Cout, int ldy, void* stream) {
    CVK_CHECK_ARG(x && w && y, "cvk_conv3x3_fwd_bf16: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout, "cvk_conv3x3_fwd_bf16: bad shape");
    CVK_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "cvk_conv3x3_fwd_bf16: Cin=%d must be a multiple of 32 (use cvk_conv3x3_fwd otherwise)", Cin);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(w), "cvk_conv3x3_fwd_bf16: x and w must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W * Cin * 4 < (1L << 31) && (long)Cout * 9 * Cin * 4 < (1L << 31), "cvk_conv3x3_fwd_bf16: tensor exceeds the 2 GiB buffer-addressing limit; split the batch");
    const int M = N * H * W, Ktot = 9 * Cin, P = cvk_cdiv(M, CVK_STAT_ROWS);
    hipStream_t s = (hipStream_t)stream;
#define CVK_BF_LAUNCH(BM_, BN_, WM_, WN_)                                                                                  \
    do {                                                                                                                  \
        const int tilesN = cvk_cdiv(ldy, BN_), tilesM = cvk_cdiv(M, BM_);                                                 \
        dim3 grid(tilesM* tilesN), block(WM_* WN_ * 64);                                                                  \
        if (stats)                                                                                                        \
            hipLaunchKernelGGL((k_conv3x3_igemm_bf16<BM_, BN_, WM_, WN_, true>), grid, block, 0, s, x, w, bias, y, stats, M, H, \
                               W, Cin, Cout, ldy, Ktot, P, tilesN);                                                       \
        else                                                                                                              \
            hipLaunchKernelGGL((k_conv3x3_igemm_bf16<BM_, BN_, WM_, WN_, false>), grid, block, 0, s, x, w, bias, y, stats, M, H, \
                               W, Cin, Cout, ldy, Ktot, P, tilesN);                                                       \
    } while (0)
    if (ldy > 64) CVK_BF_LAUNCH(128, 128, 2, 2);
    else if (ldy > 32) CVK_BF_LAUNCH(128, 64, 2, 2);
    else CVK_BF_LAUNCH(256, 32, 4, 1);
#undef CVK_BF_LAUNCH
    CVK_LAUNCH_RETURN("cvk_conv3x3_fwd_bf16");
}
