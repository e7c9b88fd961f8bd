// conv_split.hip — fp32-ACCURATE 3x3 convolution on the bf16 matrix cores (opt-in, experimental).
//
// Every fp32 operand is split exactly into three bf16 pieces  a = a1 + a2 + a3  (a1 = rn_bf16(a), a2 = rn_bf16(a - a1),
// a3 = a - a1 - a2, exact because 24 mantissa bits fit 3 x 8) on the way to LDS, and the product is rebuilt from the six
// largest cross terms  a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1  (the dropped terms are <= 2^-24 |ab|), each an exact
// bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Per product the error is ~2^-23 relative — the
// same order as one fp32 rounding — so results agree with the exact-fp32 kernels to fp32 tolerance while the matrix
// work is 6/16 of the fp32 MFMA's.  NOT the default: the headline path computes in exact fp32 (conv3x3.hip / wino.hip).
#include "conv_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// (a, b) -> three packed bf16 pairs; the residuals are exact fp32 subtractions
__device__ __forceinline__ void split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    bf16x2 h = {(__bf16)a, (__bf16)b};
    p1 = __builtin_bit_cast(unsigned, h);
    float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
    h = bf16x2{(__bf16)ra, (__bf16)rb};
    p2 = __builtin_bit_cast(unsigned, h);
    ra -= __builtin_bit_cast(float, p2 << 16);
    rb -= __builtin_bit_cast(float, p2 & 0xFFFF0000u);
    h = bf16x2{(__bf16)ra, (__bf16)rb};
    p3 = __builtin_bit_cast(unsigned, h);
}

// BKS: K slice (fp32 elements) per LDS stage, 16 or 32; Cin % BKS == 0
template <int BM, int BN, int WARPS_M, int WARPS_N, bool STATS, int BKS>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k_conv3x3_igemm_split(
    const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ Y,
    float* __restrict__ stats, int M, int H, int W, int Cin, int Cout, int ldy, int Ktot, int P, int tilesN) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int VPR = BKS / 4;          // 16-byte fp32 vectors per tile row
    constexpr int RP = NT / VPR;          // rows staged per pass
    constexpr int NA = BM / RP, NB = BN / RP;
    constexpr int PITCH = BKS + 8;        // bf16 elements per LDS row (80 B / 48 B: conflict-free ds_read_b128)
    constexpr int PLANE_A = BM * PITCH, PLANE_B = BN * PITCH;
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);
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

    const int kv = tid % VPR, r0 = tid / VPR;
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
    int lk = 0, ltap = 0, lcib = 0;

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
        lk += BKS;
        lcib += BKS;
        const int w1 = lcib >= Cin;
        lcib = w1 ? 0 : lcib;
        ltap += w1;
    };
    auto store_rows = [&](__bf16* base, int plane_stride, const f32x4& v, int row) {
        unsigned p1a, p2a, p3a, p1b, p2b, p3b;
        split2(v[0], v[1], p1a, p2a, p3a);
        split2(v[2], v[3], p1b, p2b, p3b);
        __bf16* d = base + row * PITCH + kv * 4;
        *reinterpret_cast<uint2*>(d) = uint2{p1a, p1b};
        *reinterpret_cast<uint2*>(d + plane_stride) = uint2{p2a, p2b};
        *reinterpret_cast<uint2*>(d + 2 * plane_stride) = uint2{p3a, p3b};
    };
    auto store_stage = [&](__bf16* dst, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) store_rows(dst, PLANE_A, ra[i], r0 + i * RP);
#pragma unroll
        for (int i = 0; i < NB; ++i) store_rows(dst + 3 * PLANE_A, PLANE_B, rb[i], r0 + i * RP);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int aro = (wm * TM * 32 + li) * PITCH + lh * 8;
    const int bro = 3 * PLANE_A + (wn * TN * 32 + li) * PITCH + lh * 8;
    auto mma_kg = [&](const __bf16* cur, int kg) {
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int t = 0; t < TM; ++t) a[p][t] = *reinterpret_cast<const bf16x8*>(cur + aro + p * PLANE_A + t * 32 * PITCH + kg * 16);
#pragma unroll
            for (int t = 0; t < TN; ++t) b[p][t] = *reinterpret_cast<const bf16x8*>(cur + bro + p * PLANE_B + t * 32 * PITCH + kg * 16);
        }
        // six cross terms, smallest first
        constexpr int PA_[6] = {0, 2, 1, 0, 1, 0};
        constexpr int PB_[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA_[t6]][tm], b[PB_[t6]][tn], acc[tm][tn], 0, 0, 0);
    };

    const int nK = Ktot / BKS;
    issue_loads(ra0, rb0);
    store_stage(smem, ra0, rb0);
    issue_loads(ra0, rb0);
    issue_loads(ra1, rb1);
    __syncthreads();
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + STAGE;
#define CVK_SSTEP(cur, nxt, RA, RB)                      \
    do {                                                 \
        mma_kg(cur, 0);                                  \
        store_stage(nxt, RA, RB);                        \
        issue_loads(RA, RB);                             \
        if (BKS == 32) mma_kg(cur, 1);                   \
        __syncthreads();                                 \
    } while (0)
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_SSTEP(buf0, buf1, ra0, rb0);
        CVK_SSTEP(buf1, buf0, ra1, rb1);
    }
    if (ks < nK) CVK_SSTEP(buf0, buf1, ra0, rb0);
#undef CVK_SSTEP

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

// variant: 0 = K slice 16 (two workgroups per CU), 1 = K slice 32 (one workgroup per CU)
extern "C" int cvk_conv3x3_fwd_split(const float* x, const float* w, const float* bias, float* y, float* stats, int N, int H,
                                     int W, int Cin, int Cout, int ldy, int variant, void* stream) {
    CVK_CHECK_ARG(x && w && y, "cvk_conv3x3_fwd_split: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout, "cvk_conv3x3_fwd_split: bad shape");
    CVK_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "cvk_conv3x3_fwd_split: Cin=%d must be a multiple of 32", Cin);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(w), "cvk_conv3x3_fwd_split: x and w must be 16-byte aligned");
    CVK_CHECK_ARG((long)(2 * W + 260) * Cin * 4 < (1L << 31) && (long)Cout * 9 * Cin * 4 < (1L << 31), "cvk_conv3x3_fwd_split: a tile's input window or the weight tensor exceeds the 2 GiB buffer-addressing limit");
    const int M = N * H * W, Ktot = 9 * Cin, P = cvk_cdiv(M, CVK_STAT_ROWS);
    hipStream_t s = (hipStream_t)stream;
#define CVK_SP_LAUNCH(BM_, BN_, WM_, WN_, BKS_)                                                                            \
    do {                                                                                                                  \
        const int tilesN = cvk_cdiv(ldy, BN_), tilesM = cvk_cdiv(M, BM_);                                                 \
        dim3 grid(tilesM* tilesN), block(WM_* WN_ * 64);                                                                  \
        if (stats)                                                                                                        \
            hipLaunchKernelGGL((k_conv3x3_igemm_split<BM_, BN_, WM_, WN_, true, BKS_>), grid, block, 0, s, x, w, bias, y, stats, \
                               M, H, W, Cin, Cout, ldy, Ktot, P, tilesN);                                                 \
        else                                                                                                              \
            hipLaunchKernelGGL((k_conv3x3_igemm_split<BM_, BN_, WM_, WN_, false, BKS_>), grid, block, 0, s, x, w, bias, y, stats, \
                               M, H, W, Cin, Cout, ldy, Ktot, P, tilesN);                                                 \
    } while (0)
    if (ldy > 64) { if (variant) CVK_SP_LAUNCH(128, 128, 2, 2, 32); else CVK_SP_LAUNCH(128, 128, 2, 2, 16); }
    else { if (variant) CVK_SP_LAUNCH(128, 64, 2, 2, 32); else CVK_SP_LAUNCH(128, 64, 2, 2, 16); }
#undef CVK_SP_LAUNCH
    CVK_LAUNCH_RETURN("cvk_conv3x3_fwd_split");
}
