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

    const int pb = max(m0 - W - 1, 0);    // base of this tile's input window (window_rsrc)
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)pb * Cin, (size_t)M * Cin);
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
        aoff[i] = (unsigned)(m < M ? m - pb : 0) * (unsigned)Cin * 4u + kv * 16u;
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

// ------------------------------------------------------------------------------------------------ weight-grad (bf16)
// slab[split][co][(tap,ci)] = sum_px dy[px][co] * x[px+tap][ci] on v_mfma_f32_32x32x16_bf16.  Both operands arrive
// pixel-major ([px][channel]), but the MFMA wants 8 consecutive K (= pixel) values per lane: the tiles are stored as they
// arrive, [32 px][BM + 32] bf16, and fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose: lane i of a 16-lane group
// receives column i of 4 consecutive rows).  Pitch BM + 32 elements puts the four rows of a transpose block 16 banks
// apart -> conflict-free.  Everything else (split-K slabs, two register stages, branch-free staging) as in conv3x3.hip.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p, int pitch) {
    // p: address this lane supplies for the first 4x16 block (row q, columns 4p..4p+3); second block 4 rows below
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * pitch));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k_conv3x3_wgrad_bf16(
    const float* __restrict__ X, const float* __restrict__ DY, float* __restrict__ slab, int M, int H, int W, int Cin,
    int Cout, int ld_dy, int Ktot, int chunk, int tilesN, int ntiles) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int VA = BM / 4, VB = BN / 4;
    constexpr int RPA = NT / VA, RPB = NT / VB;
    constexpr int NA = BK / RPA, NB = BK / RPB;
    constexpr int PA = BM + 32, PB = BN + 32;       // LDS row pitches (bf16 elements)
    constexpr int STAGE = BK * (PA + PB);
    static_assert(NA >= 1 && NB >= 1 && BK % RPA == 0 && BK % RPB == 0, "tile/threads mismatch");
    static_assert(((PA / 2) % 64) % 32 == 16 && ((PB / 2) % 64) % 32 == 16, "pitch must put transpose rows 16 banks apart");

    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    const int gid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int split = gid / ntiles, tile = gid - split * ntiles;
    const int c0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;
    const int mbeg = split * chunk;
    const int mend = min(M, mbeg + chunk);

    const int pbx = max(mbeg - W - 1, 0);   // operand windows start at this pixel range (window_rsrc)
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)pbx * Cin, (size_t)M * Cin);
    const __amdgpu_buffer_rsrc_t dr = window_rsrc(DY, (size_t)mbeg * ld_dy, (size_t)M * ld_dy);

    const int cva = tid % VA, pra = tid / VA;
    const int cvb = tid % VB, prb = tid / VB;
    const int coA = c0 + cva * 4;
    const bool aok = coA < Cout;
    const int colB = n0 + cvb * 4;
    const bool bok = colB < Ktot;
    const int tapB = bok ? colB / Cin : 0;
    const int ciB = colB - tapB * Cin;
    const int dyB = tapB / 3 - 1, dxB = tapB % 3 - 1;
    const unsigned shiftB = (unsigned)(((dyB * W + dxB) * Cin + ciB) * 4);
    const int HW = H * W;

    unsigned brem[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) brem[i] = (unsigned)((mbeg + prb + i * RPB) % HW);
    const FastDiv divW((unsigned)W);
    int lm = mbeg;

    f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    auto issue_loads = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = lm + pra + i * RPA;
            ra[i] = buf_load16(dr, oob_unless(aok & (m < mend), ((unsigned)(m - mbeg) * (unsigned)ld_dy + (unsigned)coA) * 4u));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = lm + prb + i * RPB;
            const int y = (int)divW.div(brem[i]);
            const int x = (int)brem[i] - y * W;
            const bool ok = bok & (m < mend) & ((unsigned)(y + dyB) < (unsigned)H) & ((unsigned)(x + dxB) < (unsigned)W);
            rb[i] = buf_load16(xr, oob_unless(ok, (unsigned)(m - pbx) * (unsigned)Cin * 4u + shiftB));
            brem[i] += BK;
            if (HW >= BK) { if (brem[i] >= (unsigned)HW) brem[i] -= (unsigned)HW; }
            else brem[i] %= (unsigned)HW;
        }
        lm += BK;
    };
    auto store_stage = [&](__bf16* dst, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            uint2 v = {pack_bf16(ra[i][0], ra[i][1]), pack_bf16(ra[i][2], ra[i][3])};
            *reinterpret_cast<uint2*>(&dst[(pra + i * RPA) * PA + cva * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            uint2 v = {pack_bf16(rb[i][0], rb[i][1]), pack_bf16(rb[i][2], rb[i][3])};
            *reinterpret_cast<uint2*>(&dst[BK * PA + (prb + i * RPB) * PB + cvb * 4]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transpose-read lane roles: 16-lane group g = lane >> 4 covers matrix rows/cols 16*(g & 1) .. +15 of a 32-wide tile
    // and K half lh = g >> 1; inside the group lane 4q + p supplies (row q, columns 4p .. 4p+3) of the 4 x 16 block
    const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
    const int aoffs = (8 * lh + tq) * PA + wm * TM * 32 + 16 * g1 + 4 * tp;
    const int boffs = BK * PA + (8 * lh + tq) * PB + wn * TN * 32 + 16 * g1 + 4 * tp;
    auto mma_kg = [&](const __bf16* cur, int kg) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = tr_frag(cur + aoffs + kg * 16 * PA + t * 32, PA);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = tr_frag(cur + boffs + kg * 16 * PB + t * 32, PB);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    };

    const int nK = (mend - mbeg + BK - 1) / BK;
    issue_loads(ra0, rb0);
    store_stage(smem, ra0, rb0);
    issue_loads(ra0, rb0);
    issue_loads(ra1, rb1);
    __syncthreads();
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + STAGE;
#define CVK_BWSTEP(cur, nxt, RA, RB)   \
    do {                               \
        mma_kg(cur, 0);                \
        store_stage(nxt, RA, RB);      \
        issue_loads(RA, RB);           \
        mma_kg(cur, 1);                \
        __syncthreads();               \
    } while (0)
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_BWSTEP(buf0, buf1, ra0, rb0);
        CVK_BWSTEP(buf1, buf0, ra1, rb1);
    }
    if (ks < nK) CVK_BWSTEP(buf0, buf1, ra0, rb0);
#undef CVK_BWSTEP

    float* out = slab + (size_t)split * Cout * Ktot;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * TN * 32 + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < Cout && col < Ktot) out[(size_t)row * Ktot + col] = acc[tm][tn][r];
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
    CVK_CHECK_ARG((long)(2 * W + 260) * Cin * 4 < (1L << 31) && (long)Cout * 9 * Cin * 4 < (1L << 31), "cvk_conv3x3_fwd_bf16: a tile's input window or the weight tensor exceeds the 2 GiB buffer-addressing limit");
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

static WgradPlan plan_wgrad_bf16(int M, int Cin_pad, int Cout) {
    WgradPlan p = plan_wgrad(M, Cin_pad, Cout);
    if (p.bn != 128) {   // the stem's 64x64 tile is not instantiated for bf16: 64x128 columns instead
        p.bn = 128;
        p.tilesN = cvk_cdiv(9 * Cin_pad, 128);
        p.splits = choose_splits(p.tilesM * p.tilesN, M);
        p.chunk = cvk_cdiv(cvk_cdiv(M, p.splits), BK) * BK;
        p.splits = cvk_cdiv(M, p.chunk);
    }
    return p;
}

extern "C" size_t cvk_conv3x3_wgrad_bf16_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_pad <= 0 || Cout <= 0) return 0;
    const WgradPlan p = plan_wgrad_bf16(N * H * W, Cin_pad, Cout);
    return (size_t)p.splits * Cout * 9 * Cin_pad * sizeof(float);
}

extern "C" int cvk_conv3x3_wgrad_bf16(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                                      int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dy && dw && workspace, "cvk_conv3x3_wgrad_bf16: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 32 && Cin_pad >= Cin, "cvk_conv3x3_wgrad_bf16: bad shape (needs Cout > 32; use cvk_conv3x3_wgrad otherwise)");
    CVK_CHECK_ARG(Cin_pad % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= Cout, "cvk_conv3x3_wgrad_bf16: Cin_pad and ld_dy must be multiples of 4, ld_dy >= Cout");
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(dy) && cvk_aligned16(workspace), "cvk_conv3x3_wgrad_bf16: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wgrad_bf16: tensor too large for 32-bit pixel indices");
    CVK_CHECK_ARG((long)H * W * W < (1L << 32), "cvk_conv3x3_wgrad_bf16: frame too large for the multiply-high row/column split");
    const int M = N * H * W, Ktot = 9 * Cin_pad;
    const WgradPlan p = plan_wgrad_bf16(M, Cin_pad, Cout);
    CVK_CHECK_ARG((long)(p.chunk + 2 * W + 2 + 2 * BK) * (Cin_pad > ld_dy ? Cin_pad : ld_dy) * 4 < (1L << 31), "cvk_conv3x3_wgrad_bf16: one pixel range exceeds the 2 GiB buffer-addressing limit");
    const size_t need = (size_t)p.splits * Cout * Ktot * sizeof(float);
    if (workspace_bytes < need) {
        cvk_set_error("cvk_conv3x3_wgrad_bf16: workspace %zu < %zu bytes", workspace_bytes, need);
        return CVK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* slab = (float*)workspace;
    dim3 grid(p.tilesM * p.tilesN * p.splits);
    if (p.bm == 128)
        hipLaunchKernelGGL((k_conv3x3_wgrad_bf16<128, 128, 2, 2>), grid, dim3(256), 0, s, x, dy, slab, M, H, W, Cin_pad, Cout, ld_dy, Ktot, p.chunk, p.tilesN, p.tilesM * p.tilesN);
    else
        hipLaunchKernelGGL((k_conv3x3_wgrad_bf16<64, 128, 2, 2>), grid, dim3(256), 0, s, x, dy, slab, M, H, W, Cin_pad, Cout, ld_dy, Ktot, p.chunk, p.tilesN, p.tilesM * p.tilesN);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cvk_set_error("cvk_conv3x3_wgrad_bf16: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    const size_t total = (size_t)Cout * 9 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(256), 0, s, slab, dw, p.splits, Cout, Cin, Cin_pad, (size_t)Cout * Ktot);
    CVK_LAUNCH_RETURN("cvk_conv3x3_wgrad_bf16");
}
