// wino.hip — 3x3 convolution as 1-D Winograd F(2,3) along the image width on the fp32 matrix cores.
//
// Same operator as conv3x3.hip (nn.Conv2d(cin,cout,3,padding=1): reference models/unet.py:11, models/segnet.py:8) with
// 1.5x fewer multiplies: per kernel row r and output-column pair (2t, 2t+1)
//     y[2t]   = m0 + m1 + m2,   y[2t+1] = m1 - m2 - m3,      m_xi = sum_{r,ci} V_xi[r][ci] * U_xi[r][ci]
//     V = (d0-d2, d1+d2, d2-d1, d1-d3)  with d_o = x[y+r-1][2t-1+o],   U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2).
// Each xi is an implicit GEMM  M_xi[tiles][Cout] = V_xi[tiles][3*Cin] * U_xi[3*Cin][Cout]  (K = 3*Cin instead of 9*Cin, and
// half as many rows).  The input transform costs nothing in memory: the staging path loads the two pixels a V element
// needs (range-checked buffer loads -> zero padding for free) and combines them with one FMA on the way to LDS.  One
// workgroup walks all four xi of its (tile block, channel block) back to back so the load pipeline never drains and
// the four accumulator flushes overlap the next xi's MFMAs.  The output transform (+bias, +BatchNorm statistics) is a
// separate HBM-bound pass (k_wino_output).  Arithmetic is exact-fp32 MFMA; only the summation order differs from
// the direct kernel (error ~1e-7 relative, same order as the direct kernel's).
#include "conv_tile.h"

namespace {

// U[xi][co][r][ci] from w[co][r][s][ci]
__global__ void k_wino_weight(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const size_t total = (size_t)Cout * 3 * Cin;
    const size_t plane = total;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t cr = i / Cin;   // co*3 + r
        const float* g = w + (cr * 3) * Cin + ci;
        const float g0 = g[0], g1 = g[Cin], g2 = g[2 * (size_t)Cin];
        U[i] = g0;
        U[plane + i] = 0.5f * (g0 + g1 + g2);
        U[2 * plane + i] = 0.5f * (g0 - g1 + g2);
        U[3 * plane + i] = g2;
    }
}

template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64, 2) void k_conv3x3_wino(
    const float* __restrict__ X, const float* __restrict__ U, float* __restrict__ Mo, int Mt, int H, int W, int Wt,
    int Cin, int Cout, int ldm, int tilesN, int Mpix, int nxi) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int RP = NT / 8;
    constexpr int NA = BM / RP, NB = BN / RP;
    constexpr int STAGE = (BM + BN) * LDT;
    static_assert(NA >= 1 && NB >= 1 && BM % RP == 0 && BN % RP == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    // 1-D grid in two regions.  Blocks [0, nfull) own one tile each and walk all four transform indices; the remaining
    // tiles (fewer than one per CU) are cut into four single-index blocks so that the last dispatch round is 4x finer
    // grained (tail loss of a 256-CU round drops from ~12 % to ~3 %).  Each region is XCD-remapped on its own; the four
    // blocks of a split tile are neighbours (same pixels -> one L2).
    const int nfull = nxi;   // (argument reused: number of full-tile blocks)
    int tile, xi_begin, xi_end;
    if ((int)blockIdx.x < nfull) {
        tile = cvk_xcd_remap(blockIdx.x, nfull);
        xi_begin = 0;
        xi_end = 4;
    } else {
        const int r = cvk_xcd_remap(blockIdx.x - nfull, gridDim.x - nfull);
        tile = nfull + (r >> 2);
        xi_begin = r & 3;
        xi_end = xi_begin + 1;
    }
    const int m0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;
    const int K3 = 3 * Cin;
    const int nK = K3 / BK;   // even by contract (Cin % 64 == 0)

    // input window (window_rsrc): starts one image row + one pixel before the first column pair of this tile
    const int pb = max((m0 / Wt) * W + 2 * (m0 % Wt) - W - 1, 0);
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)pb * Cin, (size_t)Mpix * Cin);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, 4 * Cout * K3 * 4, 0x00020000);

    const int kv = tid & 7, r0 = tid >> 3;
    // per staged row: byte offset of pixel (n, y, 2*xt) (a multiple of 256 because Cin % 64 == 0) with the 7 validity
    // bits in its low byte: bit r (r<3): image row y+r-1 exists; bit 3+o (o<4): column 2*xt-1+o exists
    unsigned arow[NA], boff[NB];
    const int HWt = H * Wt;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int t = m0 + r0 + i * RP;
        unsigned v = 0;
        if (t < Mt) {
            const int n = t / HWt, rem = t - n * HWt;
            const int y = rem / Wt, xt = rem - y * Wt;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if ((unsigned)(y + r - 1) < (unsigned)H) v |= 1u << r;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if ((unsigned)(2 * xt + o - 1) < (unsigned)W) v |= 8u << o;
            v |= (unsigned)((n * H + y) * W + 2 * xt - pb) * (unsigned)Cin * 4u;
        }
        arow[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = n0 + r0 + i * RP;
        boff[i] = co < Cout ? (unsigned)co * (unsigned)K3 * 4u + kv * 16u : OOB;
    }
    const unsigned kvb = kv * 16u;

    // Register stages: the "p" pixel of every V element is loaded two-and-a-half slices ahead (two stages), the "q"
    // pixel and the weights one slice ahead (their lines are L2-warm: q of tile xt is p of tile xt+1 / of the previous
    // xi phase, weights are re-read by every row block) -> 64 staging VGPRs, which keeps two workgroups per CU.
    f32x4 rp0[NA], rp1[NA], rq[NA], rb[NB];
    int pxi = xi_begin, pr = 0, pcib = 0;   // next slice for the p loads   (transform index, kernel row, channel base; uniform)
    int qxi = xi_begin, qr = 0, qcib = 0;   // next slice for the q / weight loads

    // xi: 0 -> d0-d2   1 -> d1+d2   2 -> d2-d1   3 -> d1-d3        (d_o sits at column 2*xt-1+o)
    auto issue_p = [&](f32x4 (&rp)[NA]) {
        const int po = pxi == 0 ? -1 : (pxi == 2 ? 1 : 0);
        const unsigned sh = (unsigned)((((pr - 1) * W + po) * Cin + pcib) * 4) + kvb;
        const unsigned need = (pxi < xi_end ? (1u << pr) : 0x80u) | (8u << (po + 1));   // 0x80 is never set: past-the-end -> zeros
#pragma unroll
        for (int i = 0; i < NA; ++i) rp[i] = buf_load16(xr, oob_unless((arow[i] & need) == need, (arow[i] & ~0xFFu) + sh));
        pcib += BK;                                  // branch-free advance (slice -> kernel row -> transform index)
        const int w1 = pcib >= Cin;
        pcib = w1 ? 0 : pcib;
        pr += w1;
        const int w2 = pr == 3;
        pr = w2 ? 0 : pr;
        pxi += w2;
    };
    float sg = 1.f;                    // V = p + sg*q for the slice whose q is in rq
    auto issue_qb = [&]() {
        const int qo = qxi == 2 ? 0 : (qxi == 3 ? 2 : 1);
        sg = qxi == 1 ? 1.f : -1.f;
        const unsigned sh = (unsigned)((((qr - 1) * W + qo) * Cin + qcib) * 4) + kvb;
        const unsigned need = (qxi < xi_end ? (1u << qr) : 0x80u) | (8u << (qo + 1));
#pragma unroll
        for (int i = 0; i < NA; ++i) rq[i] = buf_load16(xr, oob_unless((arow[i] & need) == need, (arow[i] & ~0xFFu) + sh));
        const unsigned ub = qxi < xi_end ? (unsigned)((qxi * Cout * K3 + qr * Cin + qcib) * 4) : OOB;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = buf_load16(ur, (boff[i] + ub) | ((boff[i] | ub) & OOB));
        qcib += BK;
        const int w1 = qcib >= Cin;
        qcib = w1 ? 0 : qcib;
        qr += w1;
        const int w2 = qr == 3;
        qr = w2 ? 0 : qr;
        qxi += w2;
    };
    auto store_stage = [&](float* dst, const f32x4 (&rp)[NA]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(sg, rq[i][j], rp[i][j]);
            *reinterpret_cast<f32x4*>(&dst[(r0 + i * RP) * LDT + kv * 4]) = v;
        }
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

    auto mma_kk = [&](const float* arow_, const float* brow_, int kk) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const f32x4*>(arow_ + t * 32 * LDT + kk * 8);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const f32x4*>(brow_ + t * 32 * LDT + kk * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    };

    // prologue: slice 0 into LDS stage 0; p of slices 1 and 2 and q/weights of slice 1 in flight
    issue_p(rp0);
    issue_qb();
    store_stage(smem, rp0);
    issue_qb();
    issue_p(rp0);
    issue_p(rp1);
    __syncthreads();
    const int aro = (wm * TM * 32 + li) * LDT + lh * 4;
    const int bro = BM * LDT + (wn * TN * 32 + li) * LDT + lh * 4;
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    const int rowbase = m0 + wm * TM * 32;
    const bool full = (m0 + BM <= Mt) && (n0 + BN <= ldm);

    // q / weight registers are re-issued right after the LDS store that consumed them, so they too are in flight for a
    // whole K step (an earlier version issued them at the top of the step that stores them: +20 % time in vmcnt waits)
#define CVK_WINO_STEP(cur, nxt, RP_)                  \
    do {                                              \
        mma_kk(cur + aro, cur + bro, 0);              \
        mma_kk(cur + aro, cur + bro, 1);              \
        store_stage(nxt, RP_);       /* slice ks+1 */ \
        issue_qb();                  /* slice ks+2 */ \
        issue_p(RP_);                /* slice ks+3 */ \
        mma_kk(cur + aro, cur + bro, 2);              \
        mma_kk(cur + aro, cur + bro, 3);              \
        __syncthreads();                              \
    } while (0)

    for (int xi = xi_begin; xi < xi_end; ++xi) {
        for (int ks = 0; ks < nK; ks += 2) {
            CVK_WINO_STEP(buf0, buf1, rp0);
            CVK_WINO_STEP(buf1, buf0, rp1);
        }
        // flush M_xi (the next xi's slices are already in flight / in LDS) and restart the accumulators
        float* out = Mo + (size_t)xi * Mt * ldm;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * TN * 32 + tn * 32 + li;
            if (full) {
                float* yp = out + (size_t)(rowbase + 4 * lh) * ldm + col;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        yp[(size_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * ldm] = acc[tm][tn][r];
                        acc[tm][tn][r] = 0.f;
                    }
            } else {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rowbase + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (row < Mt && col < ldm) out[(size_t)row * ldm + col] = acc[tm][tn][r];
                        acc[tm][tn][r] = 0.f;
                    }
            }
        }
    }
#undef CVK_WINO_STEP
}

// y[pixel][c] = A^T-combination of M_0..3 + bias; BatchNorm statistics partials per 64-pixel granule
// (sum and M2 about the granule mean, from bias-shifted sums).  One block = one granule x up to 1024 channels.
template <bool STATS>
__global__ __launch_bounds__(256) void k_wino_output(const float* __restrict__ Mo, int ldm, int Mt,
                                                    const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                    float* __restrict__ stats, int P, int Mpix, int H, int W, int Wt,
                                                    int C, int Cout) {
    __shared__ float red[2][1024];
    const int c0 = blockIdx.y * 1024;
    const int cw = min(1024, C - c0);
    const int cvn = cw / 4, ppp = 256 / cvn;
    const int t = threadIdx.x;
    const bool active = t < cvn * ppp;
    const int cv = t % cvn, pr = t / cvn;
    const int c = c0 + cv * 4;
    const int mbeg = blockIdx.x * CVK_STAT_ROWS, mend = min(Mpix, mbeg + CVK_STAT_ROWS);
    f32x4 sh = {0.f, 0.f, 0.f, 0.f};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        if (bias != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[j] = (c + j < Cout) ? bias[c + j] : 0.f;
        }
        const int HW = H * W;
        const size_t plane = (size_t)Mt * ldm;
        for (int m = mbeg + pr; m < mend; m += ppp) {
            const int n = m / HW, rem = m - n * HW;
            const int y = rem / W, x = rem - y * W;
            const float* p = Mo + ((size_t)(n * H + y) * Wt + (x >> 1)) * ldm + c;
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(p + plane);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(p + 2 * plane);
            f32x4 v;
            if (x & 1) {
                const f32x4 m3 = *reinterpret_cast<const f32x4*>(p + 3 * plane);
                v = m1 - m2 - m3;
            } else {
                const f32x4 m0v = *reinterpret_cast<const f32x4*>(p);
                v = m0v + m1 + m2;
            }
            v += sh;
            *reinterpret_cast<f32x4*>(Y + (size_t)m * ldy + c) = v;
            if (STATS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[j] - sh[j];
                    s1[j] += d;
                    s2[j] += d * d;
                }
            }
        }
    }
    if (!STATS) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][t * 4 + j] = s1[j];
        red[1][t * 4 + j] = s2[j];
    }
    __syncthreads();
    if (t < cvn) {
        const float cnt = (float)(mend - mbeg);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < ppp; ++p) {
                a += red[0][(p * cvn + t) * 4 + j];
                b += red[1][(p * cvn + t) * 4 + j];
            }
            const int ch = c0 + t * 4 + j;
            if (ch < Cout) {
                float m2 = b - a * a / cnt;
                stats[(size_t)blockIdx.x * Cout + ch] = a + cnt * sh[j];   // sh is per-thread: thread t owns channels c0+4t..
                stats[(size_t)(P + blockIdx.x) * Cout + ch] = m2 > 0.f ? m2 : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight-grad
// Transposed F(2,3):  dW[co][r][0..2][ci] = G^T [ (A dy) (.) (B^T d) ]  summed over column pairs, with
//   A dy = (dy0, dy0+dy1, dy0-dy1, -dy1)  (dy0/dy1 = output-gradient at columns 2t / 2t+1; we store +dy1 and flip the sign
//   in the final combination),  B^T d = the same input transform as the forward pass.  Per xi one implicit GEMM
//   P_xi[Cout][3*Cin] = E_xi^T[tiles][Cout] * V_xi[tiles][3*Cin]   (reduction over N*H*ceil(W/2) tiles: half the pixels,
//   3 instead of 9 column blocks per pixel -> 12*M*Cin*Cout FLOPs instead of 18).  Both operands are transformed on the
//   way to LDS; the 4 P_xi only meet in the (tiny, weight-sized) slab reduction, so there is no extra activation pass.
// K slices are aligned to image rows so that every coordinate is either a per-thread constant or wave-uniform (SALU):
//   W >= 64 (Wt >= 32): a slice is 32 consecutive column pairs of ONE image row (S = ceil(Wt/32) slices per row),
//   W <  64            : a slice is R = floor(32/Wt) whole image rows.
// A thread's tile inside a slice is (dr, xt) = constants; the slice contributes the uniform (row0, xbase).
// L = tiles per K slice (32, or 30 so that frame widths that are multiples of 30/60 - every CamVid level - waste nothing).
template <int BM, int BN, int WARPS_M, int WARPS_N, bool MULTIROW, int L>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64, 2) void k_wgrad_wino(
    const float* __restrict__ X, const float* __restrict__ DY, float* __restrict__ slab, int NH, int H, int W, int Wt,
    int Cin, int Cout, int ld_dy, int K3, int chunk, int tilesN, int Mpix, int ntiles, int nslices, int S, int R) {
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int TM = BM / WARPS_M / 32, TN = BN / WARPS_N / 32;
    constexpr int VA = BM / 4, VB = BN / 4;
    constexpr int RPA = NT / VA, RPB = NT / VB;
    constexpr int NA = BK / RPA, NB = BK / RPB;
    constexpr int STAGE = BK * (BM + BN);
    static_assert(NA >= 1 && NB >= 1 && BK % RPA == 0 && BK % RPB == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    // 1-D grid, XCD-remapped, pixel-range (split) major: all (tile, xi) workgroups of one pixel range are neighbours and
    // share its dy / x lines in one L2 instead of re-fetching them from HBM
    const int gid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int inner = ntiles * 4;
    const int split = gid / inner, rem_ = gid - split * inner;
    const int tile = rem_ >> 2, xi = rem_ & 3;
    const int c0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;
    const int sbeg = split * chunk;
    const int send = min(nslices, sbeg + chunk);

    const FastDiv divS((unsigned)S), divH((unsigned)H), divWt((unsigned)Wt);
    // operand windows (window_rsrc) start two image rows before the first row of this workgroup's slice range
    const int rowb = max((int)divS.div((unsigned)sbeg) * R - 2, 0);
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)rowb * W * Cin, (size_t)Mpix * Cin);
    const __amdgpu_buffer_rsrc_t dr_ = window_rsrc(DY, (size_t)rowb * W * ld_dy, (size_t)Mpix * ld_dy);

    const int cva = tid % VA, pra = tid / VA;
    const int cvb = tid % VB, prb = tid / VB;
    const int coA = c0 + cva * 4;
    const bool aok = coA < Cout;
    const int colB = n0 + cvb * 4;
    const bool bok = colB < K3;
    const int rB = bok ? colB / Cin : 0;              // kernel row of this thread's column
    const int ciB = colB - rB * Cin;
    // input-transform taps for this xi (column offsets relative to 2*xt) and signs
    const int po = xi == 0 ? -1 : (xi == 2 ? 1 : 0);
    const int qo = xi == 2 ? 0 : (xi == 3 ? 2 : 1);
    const float sgB = xi == 1 ? 1.f : -1.f;
    const float sgA = xi == 2 ? -1.f : 1.f;           // E = a0 + sgA*a1 with a0 = dy0 (off for xi 3), a1 = dy1 (off for xi 0)
    const bool useA0 = xi != 3, useA1 = xi != 0;

    // per-thread tile constants inside a slice: row delta, column pair, and the constant part of the byte offsets; the
    // slice adds wave-uniform terms only, so a load costs ~5 VALU (offset add, two range compares, predicate, OOB select)
    int adr[NA], axt[NA], bdr[NB], bxt[NB];
    unsigned aconst[NA], bconst[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int pr = pra + i * RPA;
        adr[i] = (S > 1 || R == 1) ? 0 : (int)divWt.div((unsigned)pr);
        axt[i] = pr - adr[i] * Wt;
        if ((S == 1 && adr[i] >= R) || pr >= L) axt[i] = Wt;   // beyond the slice: never valid
        aconst[i] = ((unsigned)(adr[i] * W + 2 * axt[i]) * (unsigned)ld_dy + (unsigned)coA) * 4u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pr = prb + i * RPB;
        bdr[i] = (S > 1 || R == 1) ? 0 : (int)divWt.div((unsigned)pr);
        bxt[i] = pr - bdr[i] * Wt;
        if ((S == 1 && bdr[i] >= R) || pr >= L) bxt[i] = Wt;
        bconst[i] = ((unsigned)((bdr[i] + rB - 1) * W + 2 * bxt[i]) * (unsigned)Cin + (unsigned)ciB) * 4u;
    }

    f32x4 ap0[NA], ap1[NA], bp0[NB], bp1[NB], aq[NA], bq[NB];
    int lp = sbeg, lq = sbeg;                          // next slice for the p / q loads (uniform)

    // slice -> uniform (first image row index row0 = n*H + y0, first column pair xbase)
    auto slice_origin = [&](int sl, int& row0, int& xbase, int& y0) {
        const int q = (int)divS.div((unsigned)sl);     // S == 1: q = sl
        row0 = q * R;                                  // S > 1  => R == 1
        xbase = (sl - q * S) * L;
        y0 = row0 - (int)divH.div((unsigned)row0) * H;
    };
    // which: 0 = "p" taps (dy0 / x column 2xt+po), 1 = "q" taps (dy1 / x column 2xt+qo)
    auto issue = [&](int sl, int which, f32x4 (&ar)[NA], f32x4 (&br)[NB]) {
        int row0, xbase, y0;
        slice_origin(sl, row0, xbase, y0);
        const bool live = sl < send;
        const int xlim = Wt - xbase, rlim = NH - row0;                       // uniform validity limits
        const int acol = which;                                               // dy column 2xt + which
        const unsigned sA = (unsigned)(((row0 - rowb) * W + 2 * xbase + acol) * ld_dy) * 4u;
        const bool aon = aok & live & (which ? useA1 : useA0);
        const int wlimA = W - 2 * xbase - acol;                               // need 2*axt < wlimA
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = aon & (axt[i] < xlim) & (adr[i] < rlim) & (2 * axt[i] < wlimA);
            ar[i] = buf_load16(dr_, oob_unless(ok, aconst[i] + sA));
        }
        const int o = which ? qo : po;
        const unsigned sB = (unsigned)(((row0 - rowb) * W + 2 * xbase + o) * Cin) * 4u;
        const int xoff = 2 * xbase + o;
        const bool bon = bok & live;
        const bool rowok_uniform = (unsigned)(y0 + rB - 1) < (unsigned)H;    // exact when the slice is one image row
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bool rowok = rowok_uniform;
            if (MULTIROW) {                                                   // narrow frames only (compile-time)
                const int yy = y0 + bdr[i];
                const int y = yy - (int)divH.div((unsigned)yy) * H;
                rowok = (unsigned)(y + rB - 1) < (unsigned)H;
            }
            const bool ok = bon & rowok & (bxt[i] < xlim) & (bdr[i] < rlim) & ((unsigned)(2 * bxt[i] + xoff) < (unsigned)W);
            br[i] = buf_load16(xr, oob_unless(ok, bconst[i] + sB));
        }
    };
    auto issue_p = [&](f32x4 (&ap)[NA], f32x4 (&bp)[NB]) { issue(lp, 0, ap, bp); ++lp; };
    auto issue_q = [&]() { issue(lq, 1, aq, bq); ++lq; };
    auto store_stage = [&](float* dst, const f32x4 (&ap)[NA], const f32x4 (&bp)[NB]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(sgA, aq[i][j], ap[i][j]);
            *reinterpret_cast<f32x4*>(&dst[(pra + i * RPA) * BM + cva * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(sgB, bq[i][j], bp[i][j]);
            *reinterpret_cast<f32x4*>(&dst[BK * BM + (prb + i * RPB) * BN + cvb * 4]) = v;
        }
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

    const int nK = send - sbeg;
    issue_p(ap0, bp0);
    issue_q();
    store_stage(smem, ap0, bp0);
    issue_q();
    issue_p(ap0, bp0);
    issue_p(ap1, bp1);
    __syncthreads();
    const int aco = lh * BM + wm * TM * 32 + li;
    const int bco = BK * BM + lh * BN + wn * TN * 32 + li;
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    // Imposed instruction order of a K step: four phases separated by scheduling fences, each with its own interleave —
    // MFMAs | the LDS stores of the next slice (transform FMAs, vmcnt waits) under MFMAs | the 2*(NA+NB) loads one per MFMA |
    // the rest; and a fence at the step boundary (hipcc otherwise stores at the top and loads at the bottom of the step, or
    // hoists the next step's FMAs across the barrier).  H/S/Q = phase boundaries in units of TM*TN MFMAs.
    //   masks: 0x008 MFMA, 0x020 VMEM read, 0x200 DS write
    constexpr int PH = TM * TN == 4 ? 4 : 2;                         // 16 / 4 MFMAs ahead of the stores
    constexpr int PS = TM * TN == 4 ? 8 : 5;                         // stores under 16 / 6 MFMAs
    constexpr int PQ = TM * TN == 4 ? 12 : 11;                       // loads under 16 / 12 MFMAs
#define CVK_FENCE() __builtin_amdgcn_sched_barrier(0)
#define CVK_WW_STEP(cur, nxt, AP_, BP_)                                           \
    do {                                                                          \
        mma_part(cur + aco, cur + bco, 0, PH);                                    \
        CVK_FENCE();                                                              \
        store_stage(nxt, AP_, BP_);  /* slice ks+1 */                             \
        mma_part(cur + aco, cur + bco, PH, PS);                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < NA + NB; ++q_) {                  \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x008, (PS - PH) * TM * TN / (NA + NB), 0); \
        }                                                                         \
        CVK_FENCE();                                                              \
        issue_q();                   /* slice ks+2 */                             \
        issue_p(AP_, BP_);           /* slice ks+3 */                             \
        mma_part(cur + aco, cur + bco, PS, PQ);                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 2 * (NA + NB); ++q_) {            \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
        }                                                                         \
        CVK_FENCE();                                                              \
        mma_part(cur + aco, cur + bco, PQ, L / 2);                                \
        __syncthreads();                                                          \
        CVK_FENCE();                                                              \
    } while (0)
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_WW_STEP(buf0, buf1, ap0, bp0);
        CVK_WW_STEP(buf1, buf0, ap1, bp1);
    }
    if (ks < nK) CVK_WW_STEP(buf0, buf1, ap0, bp0);
#undef CVK_WW_STEP
#undef CVK_FENCE

    float* out = slab + ((size_t)split * 4 + xi) * Cout * K3;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * TN * 32 + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < Cout && col < K3) out[(size_t)row * K3 + col] = acc[tm][tn][r];
            }
        }
}

// dw[co][r][s][ci] from the slabs: P_xi = sum over splits (fixed order), then G^T: (P0 + (P1+P2)/2, (P1-P2)/2, (P1+P2)/2 - P3)
__global__ void k_wgrad_wino_reduce(const float* __restrict__ slab, float* __restrict__ dw, int splits, int Cout, int Cin,
                                    int Cin_pad) {
    const size_t total = (size_t)Cout * 3 * Cin;
    const size_t plane = (size_t)Cout * 3 * Cin_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t cr = i / Cin;  // co*3 + r
        const float* p = slab + cr * Cin_pad + ci;
        float P[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
#pragma unroll
            for (int x = 0; x < 4; ++x) P[x] += p[((size_t)s * 4 + x) * plane];
        }
        const float h = 0.5f * (P[1] + P[2]);
        float* o = dw + (cr * 3) * Cin + ci;
        o[0] = P[0] + h;
        o[Cin] = 0.5f * (P[1] - P[2]);
        o[2 * (size_t)Cin] = h - P[3];
    }
}

struct WWPlan { int bm, tilesM, tilesN, splits, chunk, nslices, S, R, L; };
WWPlan plan_wgrad_wino(int NH, int Wt, int Cin_pad, int Cout) {
    WWPlan p;
    p.bm = Cout > 64 ? 128 : 64;
    p.tilesM = cvk_cdiv(Cout, p.bm);
    p.tilesN = cvk_cdiv(3 * Cin_pad, 128);
    // slice length: 32 tiles, or 30 when that wastes fewer MFMA K steps (useful fraction = Wt / (slices_per_row * L))
    auto useful = [&](int L) { return Wt >= L ? (double)Wt / ((double)cvk_cdiv(Wt, L) * L) : (double)((L / Wt) * Wt) / L; };
    p.L = useful(30) > useful(32) + 1e-9 ? 30 : 32;
    p.S = Wt >= p.L ? cvk_cdiv(Wt, p.L) : 1;               // slices per image row (wide frames)
    p.R = Wt >= p.L ? 1 : p.L / Wt;                        // whole image rows per slice (narrow frames)
    p.nslices = Wt >= p.L ? NH * p.S : cvk_cdiv(NH, p.R);
    const int units = p.tilesM * p.tilesN * 4;
    const int max_splits = p.nslices / 16 > 0 ? p.nslices / 16 : 1;
    int best = 1;
    double best_eff = -1.0;
    for (int s = 1; s <= max_splits && s <= 2048; ++s) {
        const long blocks = (long)units * s;
        if (blocks > 4096 && s > 1) break;
        if (blocks < 512 && s < max_splits) continue;
        const long rounds = (blocks + 255) / 256;
        const double eff = (double)blocks / (256.0 * rounds);
        if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
    }
    p.splits = best;
    p.chunk = cvk_cdiv(p.nslices, p.splits);               // slices per split
    p.splits = cvk_cdiv(p.nslices, p.chunk);
    return p;
}

}  // namespace

extern "C" int cvk_wino_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream) {
    CVK_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "cvk_wino_weight_transform: bad arguments");
    const size_t total = (size_t)Cout * 3 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wino_weight, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin);
    CVK_LAUNCH_RETURN("cvk_wino_weight_transform");
}

extern "C" size_t cvk_conv3x3_wino_workspace_bytes(int N, int H, int W, int Cout_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout_ld <= 0) return 0;
    return (size_t)4 * N * H * ((W + 1) / 2) * Cout_ld * sizeof(float);
}

extern "C" int cvk_conv3x3_wino_gemm(const float* x, const float* U, float* Mo, int N, int H, int W, int Cin, int Cout,
                                     int ldm, void* stream) {
    CVK_CHECK_ARG(x && U && Mo, "cvk_conv3x3_wino_gemm: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldm >= Cout && ldm % 4 == 0, "cvk_conv3x3_wino_gemm: bad shape");
    CVK_CHECK_ARG(Cin > 0 && Cin % 64 == 0, "cvk_conv3x3_wino_gemm: Cin=%d must be a multiple of 64 (use cvk_conv3x3_fwd otherwise)", Cin);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(U) && cvk_aligned16(Mo), "cvk_conv3x3_wino_gemm: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wino_gemm: tensor too large for 32-bit pixel indices");
    // the input may exceed 2 GiB: each workgroup addresses it through its own window (window_rsrc) of <= 256 + 2W + 4 pixels
    CVK_CHECK_ARG((long)(2 * W + 264) * Cin * 4 < (1L << 31) && (long)4 * Cout * 3 * Cin * 4 < (1L << 31), "cvk_conv3x3_wino_gemm: a tile's input window or the weight tensor exceeds the 2 GiB buffer-addressing limit");
    const int Wt = (W + 1) / 2, Mt = N * H * Wt, Mpix = N * H * W, ldy = ldm;
    hipStream_t s = (hipStream_t)stream;
    // tiles are dealt to the 256 CUs in whole rounds as four-index blocks; the remainder (and, for deep layers with few
    // tiles but long K, everything) is cut into single-index blocks
    auto full_tiles = [&](int tiles) {
        if (tiles < 1024 && Cin >= 256) return 0;
        return tiles / 256 * 256;
    };
    if (ldy > 64) {
        const int tilesN = cvk_cdiv(ldy, 128), tiles = cvk_cdiv(Mt, 128) * tilesN, nf = full_tiles(tiles);
        hipLaunchKernelGGL((k_conv3x3_wino<128, 128, 2, 2>), dim3(nf + 4 * (tiles - nf)), dim3(256), 0, s, x, U, Mo, Mt, H, W, Wt, Cin, Cout, ldy, tilesN, Mpix, nf);
    } else if (ldy > 32) {
        const int tilesN = cvk_cdiv(ldy, 64), tiles = cvk_cdiv(Mt, 128) * tilesN, nf = full_tiles(tiles);
        hipLaunchKernelGGL((k_conv3x3_wino<128, 64, 2, 2>), dim3(nf + 4 * (tiles - nf)), dim3(256), 0, s, x, U, Mo, Mt, H, W, Wt, Cin, Cout, ldy, tilesN, Mpix, nf);
    } else {   // narrow heads (e.g. the 12-class logits layer): 32-column tiles, four 32x32 wave tiles stacked in M
        const int tiles = cvk_cdiv(Mt, 128), nf = full_tiles(tiles);
        hipLaunchKernelGGL((k_conv3x3_wino<128, 32, 4, 1>), dim3(nf + 4 * (tiles - nf)), dim3(256), 0, s, x, U, Mo, Mt, H, W, Wt, Cin, Cout, ldy, 1, Mpix, nf);
    }
    CVK_LAUNCH_RETURN("cvk_conv3x3_wino_gemm");
}

extern "C" int cvk_wino_output(const float* Mo, const float* bias, float* y, float* stats, int N, int H, int W, int Cout,
                               int ldy, void* stream) {
    CVK_CHECK_ARG(Mo && y, "cvk_wino_output: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout && ldy % 4 == 0, "cvk_wino_output: bad shape");
    CVK_CHECK_ARG(cvk_aligned16(Mo) && cvk_aligned16(y), "cvk_wino_output: pointers must be 16-byte aligned");
    const int Wt = (W + 1) / 2, Mt = N * H * Wt, Mpix = N * H * W;
    const int P = cvk_cdiv(Mpix, CVK_STAT_ROWS);
    dim3 grid(P, cvk_cdiv(ldy, 1024));
    hipStream_t s = (hipStream_t)stream;
    if (stats)
        hipLaunchKernelGGL(k_wino_output<true>, grid, dim3(256), 0, s, Mo, ldy, Mt, bias, y, ldy, stats, P, Mpix, H, W, Wt, ldy, Cout);
    else
        hipLaunchKernelGGL(k_wino_output<false>, grid, dim3(256), 0, s, Mo, ldy, Mt, bias, y, ldy, stats, P, Mpix, H, W, Wt, ldy, Cout);
    CVK_LAUNCH_RETURN("cvk_wino_output");
}

extern "C" size_t cvk_conv3x3_wgrad_wino_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_pad <= 0 || Cout <= 0) return 0;
    const WWPlan p = plan_wgrad_wino(N * H, (W + 1) / 2, Cin_pad, Cout);
    return (size_t)p.splits * 4 * Cout * 3 * Cin_pad * sizeof(float);
}

extern "C" int cvk_conv3x3_wgrad_wino(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                                      int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dy && dw && workspace, "cvk_conv3x3_wgrad_wino: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin, "cvk_conv3x3_wgrad_wino: bad shape");
    CVK_CHECK_ARG(Cin_pad % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= Cout, "cvk_conv3x3_wgrad_wino: Cin_pad and ld_dy must be multiples of 4, ld_dy >= Cout");
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(dy) && cvk_aligned16(workspace), "cvk_conv3x3_wgrad_wino: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wgrad_wino: tensor too large for 32-bit pixel indices");
    const int Wt = (W + 1) / 2, Mt = N * H * Wt, Mpix = N * H * W, K3 = 3 * Cin_pad;
    CVK_CHECK_ARG((long)N * H * H < (1L << 32) && (long)N * H * Wt < (1L << 31), "cvk_conv3x3_wgrad_wino: frame too large for the multiply-high coordinate split");
    const WWPlan p = plan_wgrad_wino(N * H, Wt, Cin_pad, Cout);
    (void)Mt;
    {   // x and dy may exceed 2 GiB: a workgroup addresses only the image rows of its own slice range (window_rsrc)
        const long rows = (p.S > 1 ? p.chunk / p.S + 2 : (long)p.chunk * p.R) + 4;
        CVK_CHECK_ARG(rows * W * (Cin_pad > ld_dy ? Cin_pad : ld_dy) * 4 < (1L << 31), "cvk_conv3x3_wgrad_wino: one slice range exceeds the 2 GiB buffer-addressing limit");
    }
    const size_t need = (size_t)p.splits * 4 * Cout * K3 * sizeof(float);
    if (workspace_bytes < need) {
        cvk_set_error("cvk_conv3x3_wgrad_wino: workspace %zu < %zu bytes", workspace_bytes, need);
        return CVK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* slab = (float*)workspace;
    dim3 grid(p.tilesM * p.tilesN * p.splits * 4);
#define CVK_WW_LAUNCH(BM_, MR_, L_)                                                                                                   \
    hipLaunchKernelGGL((k_wgrad_wino<BM_, 128, 2, 2, MR_, L_>), grid, dim3(256), 0, s, x, dy, slab, N * H, H, W, Wt, Cin_pad, Cout, ld_dy, \
                       K3, p.chunk, p.tilesN, Mpix, p.tilesM * p.tilesN, p.nslices, p.S, p.R)
#define CVK_WW_PICK(BM_)                                                                        \
    do {                                                                                        \
        if (p.L == 30) { if (p.R > 1) CVK_WW_LAUNCH(BM_, true, 30); else CVK_WW_LAUNCH(BM_, false, 30); } \
        else { if (p.R > 1) CVK_WW_LAUNCH(BM_, true, 32); else CVK_WW_LAUNCH(BM_, false, 32); }           \
    } while (0)
    if (p.bm == 128) CVK_WW_PICK(128); else CVK_WW_PICK(64);
#undef CVK_WW_PICK
#undef CVK_WW_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cvk_set_error("cvk_conv3x3_wgrad_wino: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    const size_t total = (size_t)Cout * 3 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wgrad_wino_reduce, dim3(blocks), dim3(256), 0, s, slab, dw, p.splits, Cout, Cin, Cin_pad);
    CVK_LAUNCH_RETURN("cvk_conv3x3_wgrad_wino");
}
