// conv3x3.hip — 3x3 / stride 1 / pad 1 convolution on the gfx950 fp32 matrix cores.
//
// Replaces ATen's convolution forward / ConvolutionBackward behind nn.Conv2d(cin, cout, 3, padding=1)
// (reference models/unet.py:11, models/segnet.py:8; backward driven by train.py:131).
//
// Formulation: implicit GEMM, never materialising im2col.
//   forward / data-grad : C[M = N*H*W pixels][N = Cout] += A[M][K = 9*Cin] * B[K][N]
//       A row m, column k=(tap,ci) is the NHWC input pixel m shifted by `tap` (zero outside the frame): K-contiguous.
//       B is the weight matrix [Cout][9*Cin] (torch OIHW in channels_last strides): K-contiguous per output channel.
//   weight-grad         : C[M = Cout][N = 9*Cin] += A^T[K = pixels][M] * B[K][N], split over pixel ranges.
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 cycles / 4096 FLOP per SIMD), 64-wide waves,
// each wave owns 32x32 accumulator tiles.  Because this MFMA is 16x slower than bf16 the kernel is MFMA-issue bound
// with huge slack on LDS/L2, so staging is simple: 16-byte coalesced global loads -> registers (prefetched one
// K-slice ahead, in flight under the previous slice's 64 MFMAs per wave) -> padded LDS rows -> ds_read_b128 fragments.
// K order inside a slice is permuted identically for A and B (lane half h takes k = 8*kk + 4*h + j for MFMA j), which
// lets one ds_read_b128 feed four MFMAs.
#include "conv_tile.h"

namespace {

// ------------------------------------------------------------------------------------------------ forward / dgrad
// Pipeline per K slice (one barrier per slice, two LDS stages, one register stage):
//   MFMAs of kk=0  ->  registers (slice ks+1, loaded a whole slice ago) -> LDS[next]  ->  issue global loads of
//   slice ks+2 (range-checked buffer loads: out-of-frame taps / rows / columns return 0, no branches)  ->  MFMAs of
//   kk=1..3  ->  barrier.  Everything between two barriers is ONE basic block, so the compiler interleaves the ~60
//   staging instructions into the 64-cycle shadows of the 64 MFMAs instead of serialising them.
// CIN32: Cin % 32 == 0, so a K slice never straddles a filter tap and tap / shift are wave-uniform (SGPR math).
template <int BM, int BN, int WARPS_M, int WARPS_N, bool STATS, bool CIN32>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k_conv3x3_igemm(
    const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ Y,
    float* __restrict__ stats, int M, int H, int W, int Cin, int Cout, int ldy, int Ktot, int P, int tilesN) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int RP = NT / 8;  // tile rows staged per pass (8 lanes x 16 B cover one 32-float row)
    constexpr int NA = BM / RP, NB = BN / RP;
    constexpr int STAGE = (BM + BN) * LDT;
    static_assert(NA >= 1 && NB >= 1 && BM % RP == 0 && BN % RP == 0, "tile/threads mismatch");
    static_assert(!STATS || TM == 2, "BN statistics granule is 64 rows per wave");

    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    const int tile = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;

    const int pb = max(m0 - W - 1, 0);    // first pixel any tap of this tile can touch: base of the input window
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)pb * Cin, (size_t)M * Cin);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, Cout * Ktot * 4, 0x00020000);

    // ---- per-thread staging coordinates (fixed for the whole K loop)
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
        aoff[i] = (unsigned)(m < M ? m - pb : 0) * (unsigned)Cin * 4u + (CIN32 ? kv * 16u : 0u);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = n0 + r0 + i * RP;
        boff[i] = co < Cout ? (unsigned)co * (unsigned)Ktot * 4u + kv * 16u : OOB;
    }

    f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];   // two register stages: every slice is in flight for two K steps
    int lk = 0;                              // K base of the next slice to load (uniform)
    int ltap = 0, lcib = 0;                  // CIN32: its tap and channel base (uniform)
    int vtap = 0, vci = kv * 4;              // generic: this thread's (tap, ci)
    if (!CIN32) while (vci >= Cin) { vci -= Cin; ++vtap; }

    auto issue_loads = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
        if (CIN32) {
            const int t3 = (ltap * 11) >> 5;             // ltap / 3 for ltap < 32
            const int dy = t3 - 1, dx = ltap - 3 * t3 - 1;
            const unsigned sh = (unsigned)(((dy * W + dx) * Cin + lcib) * 4);
            const unsigned bit = 1u << ltap;             // tap >= 9 (past the end of K) matches no row
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = buf_load16(xr, (amask[i] & bit) ? aoff[i] + sh : OOB);
            const unsigned kb = lk < Ktot ? (unsigned)lk * 4u : OOB;
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = buf_load16(wr, (boff[i] | kb) & OOB ? OOB : boff[i] + kb);
            lcib += BK;
            if (lcib >= Cin) { lcib = 0; ++ltap; }
        } else {
            const int k = lk + kv * 4;
            const bool kok = k < Ktot;
            const int t3 = (vtap * 11) >> 5;
            const int dy = t3 - 1, dx = vtap - 3 * t3 - 1;
            const unsigned sh = (unsigned)(((dy * W + dx) * Cin + vci) * 4);
            const unsigned bit = (kok && vtap < 9) ? 1u << vtap : 0u;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = buf_load16(xr, (amask[i] & bit) ? aoff[i] + sh : OOB);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = buf_load16(wr, (kok && boff[i] != OOB) ? boff[i] + (unsigned)lk * 4u : OOB);
            vci += BK;
            while (vci >= Cin) { vci -= Cin; ++vtap; }
        }
        lk += BK;
    };
    auto store_stage = [&](float* dst, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&dst[(r0 + i * RP) * LDT + kv * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&dst[BM * LDT + (r0 + i * RP) * LDT + kv * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto mma_kk = [&](const float* arow, const float* brow, int kk) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const f32x4*>(arow + t * 32 * LDT + kk * 8);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const f32x4*>(brow + t * 32 * LDT + kk * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    };

    const int nK = (Ktot + BK - 1) / BK;
    issue_loads(ra0, rb0);            // slice 0
    store_stage(smem, ra0, rb0);
    issue_loads(ra0, rb0);            // slice 1 (zeros past the end of K: range-checked, no memory traffic)
    issue_loads(ra1, rb1);            // slice 2
    __syncthreads();
    const int aro = (wm * TM * 32 + li) * LDT + lh * 4;
    const int bro = BM * LDT + (wn * TN * 32 + li) * LDT + lh * 4;
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    // one K step: MFMAs of `cur`; meanwhile the register stage loaded two steps ago goes to `nxt` and is re-issued
    // for the slice three steps ahead.  All of it is one basic block between two barriers.
#define CVK_KSTEP(cur, nxt, RA, RB)            \
    do {                                       \
        mma_kk(cur + aro, cur + bro, 0);       \
        store_stage(nxt, RA, RB);              \
        issue_loads(RA, RB);                   \
        mma_kk(cur + aro, cur + bro, 1);       \
        mma_kk(cur + aro, cur + bro, 2);       \
        mma_kk(cur + aro, cur + bro, 3);       \
        __syncthreads();                       \
    } while (0)
    // straight-line double step (no branch inside: the compiler's vmcnt bookkeeping then leaves the newer register
    // stage in flight, `s_waitcnt vmcnt(8..15)`), odd tail peeled
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_KSTEP(buf0, buf1, ra0, rb0);
        CVK_KSTEP(buf1, buf0, ra1, rb1);
    }
    if (ks < nK) CVK_KSTEP(buf0, buf1, ra0, rb0);

    // ---- epilogue: + bias, store NHWC, fused BatchNorm statistics partials
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
    const int rowbase = m0 + wm * TM * 32;
    const bool full = (m0 + BM <= M) && (n0 + BN <= ldy) && (n0 + BN <= Cout);   // block-uniform fast path
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
            const int cnt = min(64, M - rowbase);  // rows of this wave inside the tensor (wave-uniform)
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

// ------------------------------------------------------------------------------------------------ weight-grad
// slab[split][co][k] = sum over the split's pixels of dy[m][co] * x[m + tap(k)][ci(k)]; same pipeline as above.
template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k_conv3x3_wgrad(
    const float* __restrict__ X, const float* __restrict__ DY, float* __restrict__ slab, int M, int H, int W, int Cin,
    int Cout, int ld_dy, int Ktot, int chunk, int tilesN, int ntiles) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int VA = BM / 4, VB = BN / 4;      // 16-byte vectors per staged pixel row
    constexpr int RPA = NT / VA, RPB = NT / VB;  // pixel rows staged per pass
    constexpr int NA = BK / RPA, NB = BK / RPB;
    constexpr int STAGE = BK * (BM + BN);
    static_assert(NA >= 1 && NB >= 1 && BK % RPA == 0 && BK % RPB == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];   // stage: [BK px][BM co] | [BK px][BN (tap,ci)]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    // 1-D grid, XCD-remapped, pixel-range (split) major: the tiles of one pixel range are neighbours and share its
    // dy / x lines in one L2
    const int gid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int split = gid / ntiles, tile = gid - split * ntiles;
    const int c0 = (tile / tilesN) * BM;  // first output channel of the tile
    const int n0 = (tile % tilesN) * BN;  // first (tap,ci) column of the tile
    const int mbeg = split * chunk;
    const int mend = min(M, mbeg + chunk);

    const int pbx = max(mbeg - W - 1, 0);   // operand windows start at this pixel range (see window_rsrc)
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)pbx * Cin, (size_t)M * Cin);
    const __amdgpu_buffer_rsrc_t dr = window_rsrc(DY, (size_t)mbeg * ld_dy, (size_t)M * ld_dy);

    const int cva = tid % VA, pra = tid / VA;
    const int cvb = tid % VB, prb = tid / VB;
    const int coA = c0 + cva * 4;
    const bool aok = coA < Cout;          // dy columns [Cout, ld_dy) are zero by contract (ld_dy % 4 == 0)
    const int colB = n0 + cvb * 4;
    const bool bok = colB < Ktot;
    const int tapB = bok ? colB / Cin : 0;
    const int ciB = colB - tapB * Cin;
    const int dyB = tapB / 3 - 1, dxB = tapB % 3 - 1;
    const unsigned shiftB = (unsigned)(((dyB * W + dxB) * Cin + ciB) * 4);
    const int HW = H * W;

    // Raster position (index inside its image) of this thread's B rows in the next slice to load; advanced by BK per
    // slice with one conditional wrap; row/column come from an exact multiply-high division (rem * W < 2^32).
    unsigned brem[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) brem[i] = (unsigned)((mbeg + prb + i * RPB) % HW);
    const FastDiv divW((unsigned)W);
    int lm = mbeg;                        // first pixel of the next slice to load

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
            const bool ok = bok & (m < mend) & ((unsigned)(y + dyB) < (unsigned)H) & ((unsigned)(x + dxB) < (unsigned)W);   // '&': no short-circuit branches
            rb[i] = buf_load16(xr, oob_unless(ok, (unsigned)(m - pbx) * (unsigned)Cin * 4u + shiftB));
            brem[i] += BK;
            if (HW >= BK) { if (brem[i] >= (unsigned)HW) brem[i] -= (unsigned)HW; }
            else brem[i] %= (unsigned)HW;
        }
        lm += BK;
    };
    auto store_stage = [&](float* dst, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&dst[(pra + i * RPA) * BM + cva * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&dst[BK * BM + (prb + i * RPB) * BN + cvb * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto mma_part = [&](const float* acol, const float* bcol, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = acol[2 * s * BM + t * 32];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = bcol[2 * s * BN + t * 32];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };

    const int nK = (mend - mbeg + BK - 1) / BK;
    issue_loads(ra0, rb0);
    store_stage(smem, ra0, rb0);
    issue_loads(ra0, rb0);
    issue_loads(ra1, rb1);
    __syncthreads();
    const int aco = lh * BM + wm * TM * 32 + li;
    const int bco = BK * BM + lh * BN + wn * TN * 32 + li;
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
#define CVK_WSTEP(cur, nxt, RA, RB)                 \
    do {                                            \
        mma_part(cur + aco, cur + bco, 0, 4);       \
        store_stage(nxt, RA, RB);                   \
        issue_loads(RA, RB);                        \
        mma_part(cur + aco, cur + bco, 4, 16);      \
        __syncthreads();                            \
    } while (0)
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_WSTEP(buf0, buf1, ra0, rb0);
        CVK_WSTEP(buf1, buf0, ra1, rb1);
    }
    if (ks < nK) CVK_WSTEP(buf0, buf1, ra0, rb0);

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

__global__ void k_pack_weight_fwd(const float* __restrict__ src, float* __restrict__ dst, int rows, int Cin, int Cin_pad) {
    const size_t total = (size_t)rows * Cin_pad;  // rows = Cout*9
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin_pad);
        const size_t r = i / Cin_pad;
        dst[i] = ci < Cin ? src[r * Cin + ci] : 0.f;
    }
}

// dst[ci][t][co] = src[co][8-t][ci]; 32x32 LDS transpose per tap so both sides stay coalesced
__global__ void k_pack_weight_dgrad(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int Cin_pad,
                                    int Cout_pad) {
    __shared__ float t[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int co = co0 + j, ci = ci0 + tx;
        t[j][tx] = (co < Cout && ci < Cin) ? src[((size_t)co * 9 + (8 - tap)) * Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int ci = ci0 + j, co = co0 + tx;
        if (ci < Cin_pad && co < Cout_pad) dst[((size_t)ci * 9 + tap) * Cout_pad + co] = t[tx][j];
    }
}

}  // namespace

// =================================================================================================== C ABI
extern "C" int cvk_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, float* stats, int N, int H,
                               int W, int Cin, int Cout, int ldy, void* stream) {
    CVK_CHECK_ARG(x && w && y, "cvk_conv3x3_fwd: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "cvk_conv3x3_fwd: bad shape N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
    CVK_CHECK_ARG(Cin % 4 == 0, "cvk_conv3x3_fwd: Cin=%d must be a multiple of 4 (pad on the host)", Cin);
    CVK_CHECK_ARG(ldy >= Cout, "cvk_conv3x3_fwd: ldy=%d < Cout=%d", ldy, Cout);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(w), "cvk_conv3x3_fwd: x and w must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512 && (long)H * W * Cin < (1L << 31), "cvk_conv3x3_fwd: tensor too large for 32-bit pixel indices");
    // the input may exceed 2 GiB: each workgroup addresses it through its own window (window_rsrc) of <= 256 + 2W + 2 pixels
    CVK_CHECK_ARG((long)(2 * W + 260) * Cin * 4 < (1L << 31) && (long)Cout * 9 * Cin * 4 < (1L << 31), "cvk_conv3x3_fwd: a tile's input window or the weight tensor exceeds the 2 GiB buffer-addressing limit");
    const int M = N * H * W, Ktot = 9 * Cin, P = cvk_cdiv(M, CVK_STAT_ROWS);
    const bool c32 = Cin % 32 == 0;
    hipStream_t s = (hipStream_t)stream;
#define CVK_CONV_LAUNCH(BM_, BN_, WM_, WN_)                                                                         \
    do {                                                                                                           \
        const int tilesN = cvk_cdiv(ldy, BN_), tilesM = cvk_cdiv(M, BM_);                                          \
        dim3 grid(tilesM* tilesN), block(WM_* WN_ * 64);                                                           \
        if (stats && c32)                                                                                          \
            hipLaunchKernelGGL((k_conv3x3_igemm<BM_, BN_, WM_, WN_, true, true>), grid, block, 0, s, x, w, bias, y, stats, \
                               M, H, W, Cin, Cout, ldy, Ktot, P, tilesN);                                          \
        else if (stats)                                                                                            \
            hipLaunchKernelGGL((k_conv3x3_igemm<BM_, BN_, WM_, WN_, true, false>), grid, block, 0, s, x, w, bias, y, stats, \
                               M, H, W, Cin, Cout, ldy, Ktot, P, tilesN);                                          \
        else if (c32)                                                                                              \
            hipLaunchKernelGGL((k_conv3x3_igemm<BM_, BN_, WM_, WN_, false, true>), grid, block, 0, s, x, w, bias, y, stats, \
                               M, H, W, Cin, Cout, ldy, Ktot, P, tilesN);                                          \
        else                                                                                                       \
            hipLaunchKernelGGL((k_conv3x3_igemm<BM_, BN_, WM_, WN_, false, false>), grid, block, 0, s, x, w, bias, y, stats, \
                               M, H, W, Cin, Cout, ldy, Ktot, P, tilesN);                                          \
    } while (0)
    if (ldy > 64) CVK_CONV_LAUNCH(128, 128, 2, 2);
    else if (ldy > 32) CVK_CONV_LAUNCH(128, 64, 2, 2);
    else CVK_CONV_LAUNCH(256, 32, 4, 1);
#undef CVK_CONV_LAUNCH
    CVK_LAUNCH_RETURN("cvk_conv3x3_fwd");
}

extern "C" int cvk_pack_weight_fwd(const float* w_src, float* dst, int Cout, int Cin, int Cin_pad, void* stream) {
    CVK_CHECK_ARG(w_src && dst && Cout > 0 && Cin > 0 && Cin_pad >= Cin, "cvk_pack_weight_fwd: bad arguments");
    const size_t total = (size_t)Cout * 9 * Cin_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_pack_weight_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_src, dst, Cout * 9, Cin, Cin_pad);
    CVK_LAUNCH_RETURN("cvk_pack_weight_fwd");
}

extern "C" int cvk_pack_weight_dgrad(const float* w_src, float* dst, int Cout, int Cin, int Cin_pad, int Cout_pad,
                                     void* stream) {
    CVK_CHECK_ARG(w_src && dst && Cout > 0 && Cin > 0 && Cin_pad >= Cin && Cout_pad >= Cout, "cvk_pack_weight_dgrad: bad arguments");
    dim3 grid(cvk_cdiv(Cin_pad, 32), cvk_cdiv(Cout_pad, 32), 9);
    hipLaunchKernelGGL(k_pack_weight_dgrad, grid, dim3(256), 0, (hipStream_t)stream, w_src, dst, Cout, Cin, Cin_pad, Cout_pad);
    CVK_LAUNCH_RETURN("cvk_pack_weight_dgrad");
}

// ---- weight-grad of a conv with <= 16 output channels and 64 input channels (the 12-class head, reference models/unet.py) ----
// dW[co][tap][ci] = sum_px dy[px][co] * x[px+tap][ci] is a 12 x 576 matrix reduced over 1.4 M pixels: the 32-row MFMA tiles of
// k_conv3x3_wgrad waste 62 % of the matrix pipe on padding rows and it ran at 26 TFLOP/s (0.73 ms, the most expensive
// "small" kernel of the step).  Here v_mfma_f32_16x16x4_f32: A = dy^T (16 output channels x 4 consecutive pixels of a
// row), B = x at the tap's shift (4 pixels x 16 channels) -> 36 accumulator blocks (9 taps x 4 channel blocks, 144 VGPRs).
// No LDS staging: a lane's B operand for the FOUR channel blocks of a tap is one 16-byte load (channel block q holds
// channels {4j + q}: lane j's float4 at channel 4j is its element of all four blocks), so a 4-pixel group costs 9 coalesced
// 1 KiB loads + one dy load for 36 MFMAs; the 9x re-read of x across taps/rows is served by L1/L2.
// Waves walk (image row, 128-pixel segment) units; a workgroup's four waves are added through LDS tap by tap and leave ONE
// partial slab [co][tap][ci] per workgroup; k_wgrad_reduce_small adds the slabs in a fixed order (deterministic).
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int WGS_SEG = 32;            // 4-pixel groups per unit

__global__ __launch_bounds__(256, 2) void k_wgrad_smallco(const float* __restrict__ X, const float* __restrict__ DY,
                                                         float* __restrict__ slab, int N, int H, int W, int Cout, int ld_dy) {
    constexpr int Cin = 64;
    __shared__ float red[3][16][64];   // [wave-1][q*4+i][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, k = lane >> 4;
    const int groups = (W + 3) >> 2, segs = (groups + WGS_SEG - 1) / WGS_SEG;
    const int units = N * H * segs;
    const int wg = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int nwaves = gridDim.x * 4;
    f32x4v acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int u = wg * 4 + wave; u < units; u += nwaves) {
        const int seg = u % segs, row = u / segs;          // row = n*H + y
        const int y = row % H;
        const float* const dyrow = DY + (size_t)row * W * ld_dy;
        const float* const xrow = X + (size_t)row * W * Cin + 4 * j;
        const int g0 = seg * WGS_SEG, g1 = min(groups, (seg + 1) * WGS_SEG);
        const bool rowok[3] = {y > 0, true, y + 1 < H};                          // wave-uniform
        // the ten loads of group g+1 are issued before the 36 MFMAs of group g (two waves per SIMD do not hide an L2 round trip)
        float a_cur, a_nxt = 0.f;
        f32x4v v_cur[9], v_nxt[9];
        auto load = [&](int g, float& a, f32x4v* v) {
            const int px = 4 * g + k;
            a = (j < Cout && px < W) ? dyrow[(size_t)px * ld_dy + j] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int xx = px + t % 3 - 1;
                v[t] = (rowok[t / 3] & ((unsigned)xx < (unsigned)W)) ? *reinterpret_cast<const f32x4v*>(xrow + ((long)(t / 3 - 1) * W + xx) * Cin) : zero4;
            }
        };
        load(g0, a_cur, v_cur);
        for (int g = g0; g < g1; ++g) {
            if (g + 1 < g1) load(g + 1, a_nxt, v_nxt);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, v_cur[t][q], acc[t][q], 0, 0, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < 9; ++t) v_cur[t] = v_nxt[t];
        }
    }
    // acc[tap][q][i] = dW[co = 4*k + i][tap][ci = 4*j + q]; waves 1..3 hand their sums to wave 0, tap by tap
    float* const out = slab + (size_t)blockIdx.x * Cout * 9 * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (wave > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wave - 1][q * 4 + i][lane] = acc[t][q][i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = 4 * k + i;
                f32x4v r;
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = ((acc[t][q][i] + red[0][q * 4 + i][lane]) + red[1][q * 4 + i][lane]) + red[2][q * 4 + i][lane];
                if (co < Cout) *reinterpret_cast<f32x4v*>(out + ((size_t)co * 9 + t) * Cin + 4 * j) = r;
            }
        }
        __syncthreads();
    }
}

// dw[i] = sum over the slabs, fixed order: 64 outputs x 4 slab segments per block, segments combined through LDS
__global__ __launch_bounds__(256) void k_wgrad_reduce_small(const float* __restrict__ slab, float* __restrict__ dw, int splits, int n) {
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    const int per = (splits + 3) / 4, s0 = sg * per, s1 = min(splits, s0 + per);
    float a = 0.f, b = 0.f;
    if (i < n) {
        int s = s0;
        for (; s + 2 <= s1; s += 2) { a += slab[(size_t)s * n + i]; b += slab[(size_t)(s + 1) * n + i]; }
        if (s < s1) a += slab[(size_t)s * n + i];
    }
    part[sg][o] = a + b;
    __syncthreads();
    if (sg == 0 && i < n) dw[i] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
}

constexpr int WGS_BLOCKS = 512;        // two workgroups per CU

static inline bool wgrad_smallco(int Cin, int Cin_pad, int Cout) { return Cout <= 16 && Cin == 64 && Cin_pad == 64; }

extern "C" size_t cvk_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_pad <= 0 || Cout <= 0) return 0;
    const WgradPlan p = plan_wgrad(N * H * W, Cin_pad, Cout);
    const size_t need = (size_t)p.splits * Cout * 9 * Cin_pad * sizeof(float);
    const size_t small = wgrad_smallco(Cin_pad, Cin_pad, Cout) ? (size_t)WGS_BLOCKS * Cout * 9 * Cin_pad * sizeof(float) : 0;
    return need > small ? need : small;
}

extern "C" int cvk_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                                 int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dy && dw && workspace, "cvk_conv3x3_wgrad: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin, "cvk_conv3x3_wgrad: bad shape");
    CVK_CHECK_ARG(Cin_pad % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= Cout, "cvk_conv3x3_wgrad: Cin_pad=%d and ld_dy=%d must be multiples of 4, ld_dy >= Cout", Cin_pad, ld_dy);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(dy) && cvk_aligned16(workspace), "cvk_conv3x3_wgrad: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wgrad: tensor too large for 32-bit pixel indices");
    CVK_CHECK_ARG((long)H * W * W < (1L << 32), "cvk_conv3x3_wgrad: frame too large for the multiply-high row/column split");
    const int M = N * H * W, Ktot = 9 * Cin_pad;
    if (wgrad_smallco(Cin, Cin_pad, Cout) && (long)M * 64 < (1L << 40)) {
        const size_t need_s = (size_t)WGS_BLOCKS * Cout * 9 * Cin_pad * sizeof(float);
        if (workspace_bytes < need_s) {
            cvk_set_error("cvk_conv3x3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need_s);
            return CVK_EWORKSPACE;
        }
        hipStream_t s = (hipStream_t)stream;
        const int units = N * H * cvk_cdiv(cvk_cdiv(W, 4), WGS_SEG);
        const int blocks = units < WGS_BLOCKS * 4 ? cvk_cdiv(units, 4) : WGS_BLOCKS;
        hipLaunchKernelGGL(k_wgrad_smallco, dim3(blocks), dim3(256), 0, s, x, dy, (float*)workspace, N, H, W, Cout, ld_dy);
        const int n = Cout * 9 * Cin;
        hipLaunchKernelGGL(k_wgrad_reduce_small, dim3(cvk_cdiv(n, 64)), dim3(256), 0, s, (const float*)workspace, dw, blocks, n);
        CVK_LAUNCH_RETURN("cvk_conv3x3_wgrad");
    }
    const WgradPlan p = plan_wgrad(M, Cin_pad, Cout);
    // x and dy may exceed 2 GiB: a workgroup addresses only its own pixel range (window_rsrc)
    CVK_CHECK_ARG((long)(p.chunk + 2 * W + 2 + 2 * BK) * (Cin_pad > ld_dy ? Cin_pad : ld_dy) * 4 < (1L << 31), "cvk_conv3x3_wgrad: one pixel range exceeds the 2 GiB buffer-addressing limit");
    const size_t need = (size_t)p.splits * Cout * Ktot * sizeof(float);
    if (workspace_bytes < need) {
        cvk_set_error("cvk_conv3x3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
        return CVK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* slab = (float*)workspace;
    dim3 grid(p.tilesM * p.tilesN * p.splits);
    // dy columns [Cout, ld_dy) are zero by contract, so the last 16-byte vector of a row may straddle Cout
    if (p.bm == 128)
        hipLaunchKernelGGL((k_conv3x3_wgrad<128, 128, 2, 2>), grid, dim3(256), 0, s, x, dy, slab, M, H, W, Cin_pad, Cout, ld_dy, Ktot, p.chunk, p.tilesN, p.tilesM * p.tilesN);
    else if (p.bm == 64 && p.bn == 64)
        hipLaunchKernelGGL((k_conv3x3_wgrad<64, 64, 2, 2>), grid, dim3(256), 0, s, x, dy, slab, M, H, W, Cin_pad, Cout, ld_dy, Ktot, p.chunk, p.tilesN, p.tilesM * p.tilesN);
    else if (p.bm == 64)
        hipLaunchKernelGGL((k_conv3x3_wgrad<64, 128, 2, 2>), grid, dim3(256), 0, s, x, dy, slab, M, H, W, Cin_pad, Cout, ld_dy, Ktot, p.chunk, p.tilesN, p.tilesM * p.tilesN);
    else
        hipLaunchKernelGGL((k_conv3x3_wgrad<32, 256, 1, 4>), grid, dim3(256), 0, s, x, dy, slab, M, H, W, Cin_pad, Cout, ld_dy, Ktot, p.chunk, p.tilesN, p.tilesM * p.tilesN);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cvk_set_error("cvk_conv3x3_wgrad: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    const size_t total = (size_t)Cout * 9 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(256), 0, s, slab, dw, p.splits, Cout, Cin, Cin_pad, (size_t)Cout * Ktot);
    CVK_LAUNCH_RETURN("cvk_conv3x3_wgrad");
}
