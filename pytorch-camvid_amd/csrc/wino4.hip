// wino4.hip — 3x3 convolution as 1-D Winograd F(4,3) along the image width on the fp32 matrix cores.
//
// Same operator as conv3x3.hip / wino.hip (nn.Conv2d(cin,cout,3,padding=1): reference models/unet.py:11,
// models/segnet.py:8) with 2x fewer multiplies than the direct form (F(2,3) in wino.hip: 1.5x): per kernel row r and
// group of four output columns 4t..4t+3, six products instead of twelve
//     m_xi = sum_{r,ci} V_xi[r][ci] * U_xi[r][ci],   xi = 0..5,      d_j = x[y+r-1][4t-1+j], j = 0..5
//     V = B^T d:  V0 = 4d0 - 5d2 + d4            U = G g:  U0 = g0/4
//                 V1 = -4d1 - 4d2 + d3 + d4                U1 = -(g0 + g1 + g2)/6
//                 V2 =  4d1 - 4d2 - d3 + d4                U2 = -(g0 - g1 + g2)/6
//                 V3 = -2d1 -  d2 + 2d3 + d4               U3 = g0/24 + g1/12 + g2/6
//                 V4 =  2d1 -  d2 - 2d3 + d4               U4 = g0/24 - g1/12 + g2/6
//                 V5 =  4d1 - 5d3 + d5                     U5 = g2
//     y[4t]   = m0 + m1 + m2 + m3 + m4          y[4t+1] = (m1 - m2) + 2(m3 - m4)
//     y[4t+2] = (m1 + m2) + 4(m3 + m4)          y[4t+3] = (m1 - m2) + 8(m3 - m4) + m5
// (interpolation points 0, +-1, +-2, inf).  Each xi is an implicit GEMM M_xi[N*H*ceil(W/4)][Cout] = V_xi[.][3*Cin] * U_xi:
// 6 GEMM units per 4 output columns = 9*M*Cin*Cout executed FLOPs instead of 18.  As in wino.hip the input transform
// never touches HBM: the staging path loads the (up to four) pixels a V element needs with range-checked buffer loads
// (zero padding for free) and combines them with three FMAs on the way to LDS; one workgroup walks the six xi of its
// tile back to back.  Arithmetic is exact-fp32 MFMA; the transform constants (4, 5, 8) cost ~2.5x the rounding error of
// F(2,3) (measured 6e-7 relative rms against fp64 for K = 768, direct fp32 summation: 4e-7).
#include "conv_tile.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

// U[xi][co][r][ci] from w[co][r][s][ci]; evaluated in double, rounded once
__global__ void k_wino4_weight(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const size_t total = (size_t)Cout * 3 * Cin;
    const size_t plane = total;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t cr = i / Cin;   // co*3 + r
        const float* g = w + (cr * 3) * Cin + ci;
        const double g0 = g[0], g1 = g[Cin], g2 = g[2 * (size_t)Cin];
        U[i] = (float)(0.25 * g0);
        U[plane + i] = (float)(-(g0 + g1 + g2) / 6.0);
        U[2 * plane + i] = (float)(-(g0 - g1 + g2) / 6.0);
        U[3 * plane + i] = (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0);
        U[4 * plane + i] = (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0);
        U[5 * plane + i] = (float)g2;
    }
}

// Data-grad weights in one step: U_d[xi][ci][r][co] = (G g)_xi with g_s = w[co][2-r][2-s][ci] (the 180-degree-rotated,
// channel-transposed filter cvk_pack_weight_dgrad builds), straight from w — 32x32 LDS transposes keep both sides coalesced.
__global__ __launch_bounds__(256) void k_wino4_weight_dgrad(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    __shared__ float t[3][32][33];
    const int r = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int co = co0 + j, ci = ci0 + tx;
        const bool ok = co < Cout && ci < Cin;
        const float* g = w + (((size_t)co * 3 + (2 - r)) * 3) * Cin + ci;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) t[s_][j][tx] = ok ? g[(size_t)s_ * Cin] : 0.f;
    }
    __syncthreads();
    const size_t plane = (size_t)Cin * 3 * Cout;
    for (int j = ty; j < 32; j += 8) {
        const int ci = ci0 + j, co = co0 + tx;
        if (ci >= Cin || co >= Cout) continue;
        const double g0 = t[2][tx][j], g1 = t[1][tx][j], g2 = t[0][tx][j];
        float* o = U + ((size_t)ci * 3 + r) * Cout + co;
        o[0] = (float)(0.25 * g0);
        o[plane] = (float)(-(g0 + g1 + g2) / 6.0);
        o[2 * plane] = (float)(-(g0 - g1 + g2) / 6.0);
        o[3 * plane] = (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0);
        o[4 * plane] = (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0);
        o[5 * plane] = (float)g2;
    }
}

template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64, 2) void k_conv3x3_wino4(
    const float* __restrict__ X, const float* __restrict__ U, float* __restrict__ Mo, int Mt, int H, int W, int Wt,
    int Cin, int Cout, int ldm, int tilesN, int Mpix, int nfull, int ksplit) {
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

    // 1-D grid in two regions (as in wino.hip): blocks [0, nfull) own one tile each and walk all six transform indices;
    // the remaining tiles are cut into six single-index blocks (finer last dispatch round), and each of those into
    // `ksplit` blocks over equal parts of the K = 3*Cin reduction (layers with few tiles: the partial planes M[kp][xi]
    // are summed by k_wino4_output).  The 6*ksplit blocks of a tile are neighbours (same pixels -> one L2).
    int tile, xi_begin, xi_end, kp = 0;
    if ((int)blockIdx.x < nfull) {
        tile = cvk_xcd_remap(blockIdx.x, nfull);
        xi_begin = 0;
        xi_end = 6;
    } else {
        const int r = cvk_xcd_remap(blockIdx.x - nfull, gridDim.x - nfull);
        const int per = 6 * ksplit;
        const int q = r / per, rem = r - q * per;
        tile = nfull + q;
        xi_begin = rem / ksplit;
        kp = rem - xi_begin * ksplit;
        xi_end = xi_begin + 1;
    }
    const int m0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;
    const int K3 = 3 * Cin;
    const int nK = K3 / BK / ksplit;   // K steps of this block per transform index; even by contract

    // input window (window_rsrc): starts one image row + one pixel before the first column group of this tile
    const int pb = max((m0 / Wt) * W + 4 * (m0 % Wt) - W - 1, 0);
    const __amdgpu_buffer_rsrc_t xr = window_rsrc(X, (size_t)pb * Cin, (size_t)Mpix * Cin);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, 6 * Cout * K3 * 4, 0x00020000);

    const int kv = tid & 7, r0 = tid >> 3;
    // per staged row: byte offset of pixel (n, y, 4*xt) = d1 (a multiple of 256 because Cin % 64 == 0) with 8 validity bits
    // in its low byte: bit r (r<3): image row y+r-1 exists; bits 3..7: columns d0, d2, d3, d4, d5 exist (d1 always does)
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
            if (xt > 0) v |= 8u;
#pragma unroll
            for (int j = 2; j < 6; ++j)
                if (4 * xt + j - 1 < W) v |= 8u << (j - 1);
            v |= (unsigned)((n * H + y) * W + 4 * xt - pb) * (unsigned)Cin * 4u;
        } else {
            v = OOB;   // no flags: every load of this row is out of range
        }
        arow[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = n0 + r0 + i * RP;
        boff[i] = co < Cout ? (unsigned)co * (unsigned)K3 * 4u + kv * 16u : OOB;
    }
    const unsigned kvb = kv * 16u;

    // One register stage: the (up to) four pixels of every V element and the weights are issued right after the LDS store
    // that consumed the previous slice, i.e. they are in flight for one whole K step.
    f32x4 ra0[NA], ra1[NA], ra2[NA], ra3[NA], rb[NB];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;              // V = c0*a0 + c1*a1 + c2*a2 + a3 for the slice held in ra*
    // next slice to load (transform index, kernel row, channel base; uniform) and the number of live slices left
    const int k0 = kp * nK * BK;                     // first K element of this block's part
    int lxi = xi_begin, lr = k0 / Cin, lcib = k0 - (k0 / Cin) * Cin;
    int lleft = (xi_end - xi_begin) * nK;

    // On this part the fp32 MFMAs and the vector ALU share issue/execute bandwidth: every VALU instruction in the K loop
    // costs ~3.5 cycles of matrix time (tools/micro/mfma_peak.hip: a bare MFMA loop reaches 98 % of peak, +128 VALU per 64
    // MFMAs 73 %).  So the per-load address/validity arithmetic is hoisted out of the slice loop: all Cin/32 slices of one
    // (transform index, kernel row) group read the same pixels and differ only in the channel base, which travels in the
    // SGPR offset of the buffer load.  aoff[tap][row] (byte offset, bit 31 set = out of frame -> the load returns 0) is
    // recomputed once per group; per-slice constants come from nibble tables indexed by the uniform transform index:
    //   nibble xi of TJ0/TJ1/TJ2 = columns j of the first three taps (d_j), TC0/TC1/TC2 = their coefficients + 8;
    //   bit xi of FOUR: a fourth tap d4 with coefficient 1 (xi 1..4); byte j of COLBIT: the validity flag of column d_j.
    constexpr unsigned TJ0 = 0x00111110u, TJ1 = 0x00322222u, TJ2 = 0x00533334u;
    constexpr unsigned TC0 = 0x00CA6C4Cu, TC1 = 0x00377443u, TC2 = 0x0096A799u, FOUR = 0x1Eu;
    constexpr unsigned long long COLBIT = 0x0000804020100008ULL;
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0, 0x00020000);   // every load -> 0
    unsigned aoff0[NA], aoff1[NA], aoff2[NA], aoff3[NA];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;              // coefficients of the group being issued (c0..c2: of the slice in flight)
    auto regroup = [&]() {                                            // (lxi, lr) changed: taps, coefficients, offsets
        const unsigned sh4 = 4u * (unsigned)lxi;                      // lxi <= 7
        const unsigned j0 = (TJ0 >> sh4) & 15u, j1 = (TJ1 >> sh4) & 15u, j2 = (TJ2 >> sh4) & 15u;
        g0 = (float)((int)((TC0 >> sh4) & 15u) - 8);
        g1 = (float)((int)((TC1 >> sh4) & 15u) - 8);
        g2 = (float)((int)((TC2 >> sh4) & 15u) - 8);
        const unsigned four = (FOUR >> lxi) & 1u;
        const unsigned rowbit = 1u << lr;
        const unsigned base = (unsigned)((((lr - 1) * W - 1) * Cin) * 4) + kvb;   // column d0, channel 0
        const unsigned cs = (unsigned)Cin * 4u;
        const unsigned s0 = base + j0 * cs, s1 = base + j1 * cs, s2 = base + j2 * cs, s3 = base + 4 * cs;
        const unsigned n0_ = rowbit | (unsigned)((COLBIT >> (8u * j0)) & 0xFFu);
        const unsigned n1_ = rowbit | (unsigned)((COLBIT >> (8u * j1)) & 0xFFu);
        const unsigned n2_ = rowbit | (unsigned)((COLBIT >> (8u * j2)) & 0xFFu);
        const unsigned n3_ = (rowbit | 0x40u) | (four - 1u);           // no fourth tap: never satisfied -> zero
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned a = arow[i], o = a & ~0xFFu;
            aoff0[i] = oob_unless((a & n0_) == n0_, o + s0);
            aoff1[i] = oob_unless((a & n1_) == n1_, o + s1);
            aoff2[i] = oob_unless((a & n2_) == n2_, o + s2);
            aoff3[i] = oob_unless((a & n3_) == n3_, o + s3);
        }
    };
    auto issue = [&]() {
        const bool live = lleft > 0;                                   // past the last slice: zeros, no memory access
        --lleft;
        c0 = g0; c1 = g1; c2 = g2;                                     // used by the LDS store of this slice, one step later
        const __amdgpu_buffer_rsrc_t xs = live ? xr : null_rsrc;
        const __amdgpu_buffer_rsrc_t us = live ? ur : null_rsrc;
        const unsigned sa = (unsigned)lcib * 4u;                       // SGPR offsets: the only per-slice address terms
        const unsigned sb = (unsigned)((lxi * Cout * K3 + lr * Cin + lcib) * 4);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, aoff0[i], sa, 0));
            ra1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, aoff1[i], sa, 0));
            ra2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, aoff2[i], sa, 0));
            ra3[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, aoff3[i], sa, 0));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(us, boff[i], sb, 0));
        lcib += BK;                                  // uniform advance (slice -> kernel row -> transform index)
        const int w1 = lcib >= Cin;
        lcib = w1 ? 0 : lcib;
        lr += w1;
        const int w2 = lr == 3;
        lr = w2 ? 0 : lr;
        lxi += w2;
    };
    auto store_stage = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(c0, ra0[i][j], fmaf(c1, ra1[i][j], fmaf(c2, ra2[i][j], ra3[i][j])));
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

    // prologue: slice 0 into LDS stage 0; slice 1 in flight (both in the first group: groups hold an even number of slices)
    regroup();
    issue();
    store_stage(smem);
    issue();
    __syncthreads();
    const int aro = (wm * TM * 32 + li) * LDT + lh * 4;
    const int bro = BM * LDT + (wn * TN * 32 + li) * LDT + lh * 4;
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    const int rowbase = m0 + wm * TM * 32;
    const bool full = (m0 + BM <= Mt) && (n0 + BN <= ldm);

    // Instruction order inside a K step, imposed on the scheduler (left alone, hipcc sinks all loads of the step below its
    // MFMAs and consumes them at the top of the next step: a quarter of a step in flight).  16 MFMAs; the 8 LDS stores of
    // the next slice (each waits for its loads, issued a whole step earlier) under 16 MFMAs; the 4*NA+NB loads of the slice
    // after that, one per MFMA; the remaining MFMAs.   masks: 0x008 MFMA, 0x020 VMEM read, 0x200 DS write
#define CVK_WINO4_PIPELINE()                                                  \
    if (TM * TN == 4) {              /* 64 MFMAs per step */                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                   \
        _Pragma("unroll") for (int q_ = 0; q_ < NA + NB; ++q_) {              \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                \
            __builtin_amdgcn_sched_group_barrier(0x008, 16 / (NA + NB), 0);   \
        }                                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < 4 * NA + NB; ++q_) {          \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
        }                                                                     \
    } else if (TM * TN == 2) {       /* 32 MFMAs per step (64-column tile): 4 | 6 stores | 18 loads | 4 */ \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                    \
        _Pragma("unroll") for (int q_ = 0; q_ < NA + NB; ++q_) {              \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
        }                                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < 4 * NA + NB; ++q_) {          \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
        }                                                                     \
    }
#define CVK_WINO4_STEP(cur, nxt)                      \
    do {                                              \
        mma_kk(cur + aro, cur + bro, 0);              \
        mma_kk(cur + aro, cur + bro, 1);              \
        store_stage(nxt);            /* slice ks+1 */ \
        issue();                     /* slice ks+2 */ \
        mma_kk(cur + aro, cur + bro, 2);              \
        mma_kk(cur + aro, cur + bro, 3);              \
        CVK_WINO4_PIPELINE();                         \
        __syncthreads();                              \
        __builtin_amdgcn_sched_barrier(0);   /* nothing crosses the step boundary: hipcc otherwise hoists the next step's \
                                                transform FMAs up here, i.e. consumes the loads right after issuing them */ \
    } while (0)

    for (int xi = xi_begin; xi < xi_end; ++xi) {
        for (int ks = 0; ks < nK; ks += 2) {
            if (lcib == 0) regroup();        // the two slices this pair issues open a new (transform index, kernel row) group
            CVK_WINO4_STEP(buf0, buf1);
            CVK_WINO4_STEP(buf1, buf0);
        }
        // flush M_xi (the next xi's first slice is already in LDS) and restart the accumulators
        float* out = Mo + (size_t)(kp * 6 + xi) * Mt * ldm;
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
#undef CVK_WINO4_STEP
#undef CVK_WINO4_PIPELINE
}

// y[pixel][c] = A^T-combination of M_0..5 + bias; BatchNorm statistics partials per 64-pixel granule (as k_wino_output).
template <bool STATS>
__global__ __launch_bounds__(256) void k_wino4_output(const float* __restrict__ Mo, int ldm, int Mt,
                                                     const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                     float* __restrict__ stats, int P, int Mpix, int H, int W, int Wt,
                                                     int C, int Cout, int ksplit, int chunk) {
    __shared__ float red[2][1024];
    const int c0 = blockIdx.y * chunk;              // chunk <= 1024 channels per block (256 for small layers: more blocks)
    const int cw = min(chunk, C - c0);
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
            const float* p = Mo + ((size_t)(n * H + y) * Wt + (x >> 2)) * ldm + c;
            const int i = x & 3;
            // plane xi of this pixel's column group, summed over the K parts (fixed order)
            auto mplane = [&](int xi) {
                f32x4 a = *reinterpret_cast<const f32x4*>(p + (size_t)xi * plane);
                for (int k = 1; k < ksplit; ++k) a += *reinterpret_cast<const f32x4*>(p + (size_t)(k * 6 + xi) * plane);
                return a;
            };
            const f32x4 m1 = mplane(1), m2 = mplane(2), m3 = mplane(3), m4 = mplane(4);
            f32x4 v;
            if (i == 0) {
                v = mplane(0) + (m1 + m2) + (m3 + m4);
            } else if (i == 1) {
                v = (m1 - m2) + 2.f * (m3 - m4);
            } else if (i == 2) {
                v = (m1 + m2) + 4.f * (m3 + m4);
            } else {
                v = (m1 - m2) + 8.f * (m3 - m4) + mplane(5);
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
                stats[(size_t)blockIdx.x * Cout + ch] = a + cnt * sh[j];
                stats[(size_t)(P + blockIdx.x) * Cout + ch] = m2 > 0.f ? m2 : 0.f;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ weight-grad
// Transposed F(4,3):  dW[co][r][0..2][ci] = G^T [ (A dy) (.) (B^T d) ]  summed over groups of four columns, with
//   E = A dy:  E0 = dy0, E1 = dy0+dy1+dy2+dy3, E2 = dy0-dy1+dy2-dy3, E3 = dy0+2dy1+4dy2+8dy3, E4 = dy0-2dy1+4dy2-8dy3, E5 = dy3
//   (dy_i = output gradient at column 4t+i, zero beyond the right edge),  V = B^T d as in the forward kernel, and per xi one
//   implicit GEMM  P_xi[Cout][3*Cin] = E_xi^T[tiles][Cout] * V_xi[tiles][3*Cin]  over N*H*ceil(W/4) tiles: 6 GEMM units per
//   4 columns = 9*M*Cin*Cout executed FLOPs instead of 18 (wino.hip's transposed F(2,3): 12).
// Transforming both operands on the fly would need 8 taps per K element pair (128 staging VGPRs); instead E1..E4 are
// written once by k_wino4_dy_transform (an HBM pass of 2x the dy bytes) and read back as plain rows, E0/E5 are columns of
// dy itself, and only V is transformed in the staging path (four taps, three FMAs).  Slices, grid and slab reduction as
// in wino.hip's k_wgrad_wino.
__global__ void k_wino4_dy_transform(const float* __restrict__ DY, int ld, float* __restrict__ E, int NH, int W, int Wt) {
    const int cvn = ld / 4;
    const size_t total = (size_t)NH * Wt * cvn;
    const size_t plane = (size_t)NH * Wt * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        const size_t t = i / cvn;                       // row * Wt + xt
        const int xt = (int)(t % Wt);
        const size_t row = t / Wt;
        const float* p = DY + (row * W + 4 * (size_t)xt) * ld + cv * 4;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(p);
        const f32x4 d1 = 4 * xt + 1 < W ? *reinterpret_cast<const f32x4*>(p + ld) : z;
        const f32x4 d2 = 4 * xt + 2 < W ? *reinterpret_cast<const f32x4*>(p + 2 * ld) : z;
        const f32x4 d3 = 4 * xt + 3 < W ? *reinterpret_cast<const f32x4*>(p + 3 * ld) : z;
        float* o = E + t * ld + cv * 4;
        const f32x4 a = d0 + d2, b = d1 + d3, c = d0 + 4.f * d2, d = 2.f * d1 + 8.f * d3;
        *reinterpret_cast<f32x4*>(o) = a + b;
        *reinterpret_cast<f32x4*>(o + plane) = a - b;
        *reinterpret_cast<f32x4*>(o + 2 * plane) = c + d;
        *reinterpret_cast<f32x4*>(o + 3 * plane) = c - d;
    }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool MULTIROW, int L>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64, 2) void k_wgrad_wino4(
    const float* __restrict__ X, const float* __restrict__ DY, const float* __restrict__ E, float* __restrict__ slab, int NH,
    int H, int W, int Wt, int Cin, int Cout, int ld_dy, int K3, int chunk, int tilesN, int Mpix, int ntiles, int nslices,
    int S, int R) {
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

    // 1-D grid, XCD-remapped, pixel-range (split) major: the (tile, xi) workgroups of one range share its lines in one L2
    const int gid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int inner = ntiles * 6;
    const int split = gid / inner, rem_ = gid - split * inner;
    const int tile = rem_ / 6, xi = rem_ - tile * 6;
    const int c0 = (tile / tilesN) * BM;
    const int n0 = (tile % tilesN) * BN;
    const int sbeg = split * chunk;
    const int send = min(nslices, sbeg + chunk);

    const FastDiv divS((unsigned)S), divH((unsigned)H), divWt((unsigned)Wt);
    // operand windows (window_rsrc) start two image rows before the first row of this workgroup's slice range
    const int rowb = max((int)divS.div((unsigned)sbeg) * R - 2, 0);
    // E operand: xi 0 / 5 are columns 0 / 3 of the dy groups; xi 1..4 are rows of the transformed planes E[xi-1]
    const bool from_dy = (xi == 0) | (xi == 5);
    const int acol = xi == 5 ? 3 : 0;
    const int RS = from_dy ? W : Wt, XS = from_dy ? 4 : 1;          // row / tile strides of the A operand, in pixels
    const size_t Mt = (size_t)NH * Wt;
    // X window: its base sits one image row + one pixel BEFORE row `rowb` (possibly before the tensor: only valid taps are
    // ever dereferenced), so that a tap's offset splits into two non-negative parts — a per-thread VGPR part
    // ((bdr + rB)*W + 4*bxt) and a slice-uniform SGPR part ((row0 - rowb)*W + 4*xbase + j); the buffer range check sees
    // the VGPR part only.
    const long xfirst = ((long)rowb * W - (W + 1)) * Cin;
    const size_t xbytes = ((size_t)((long)Mpix * Cin - xfirst)) * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(X + xfirst), 0, (int)(xbytes < 0x7FFFFFFFu ? xbytes : 0x7FFFFFFFu), 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = from_dy ? window_rsrc(DY, (size_t)rowb * W * ld_dy, (size_t)Mpix * ld_dy)
                                              : window_rsrc(E + (size_t)(xi - 1) * Mt * ld_dy, (size_t)rowb * Wt * ld_dy, Mt * ld_dy);

    const int cva = tid % VA, pra = tid / VA;
    const int cvb = tid % VB, prb = tid / VB;
    const int coA = c0 + cva * 4;
    const bool aok = coA < Cout;
    const int colB = n0 + cvb * 4;
    const bool bok = colB < K3;
    const int rB = bok ? colB / Cin : 0;              // kernel row of this thread's column
    const int ciB = colB - rB * Cin;
    // input-transform taps of this xi: columns d_j (x column 4*xt - 1 + j) and coefficients; fourth tap d4 (coefficient 1) for xi 1..4
    const int j0 = xi == 0 ? 0 : 1;
    const int j1 = xi == 5 ? 3 : 2;
    const int j2 = xi == 0 ? 4 : (xi == 5 ? 5 : 3);
    const bool four = (unsigned)(xi - 1) < 4u;
    const float c0f = (float)((int)((0x00CA6C4Cu >> (4 * xi)) & 15u) - 8);
    const float c1f = (float)((int)((0x00377443u >> (4 * xi)) & 15u) - 8);
    const float c2f = (float)((int)((0x0096A799u >> (4 * xi)) & 15u) - 8);

    // per-thread tile constants inside a slice: row delta, column group, and the constant part of the byte offsets
    int adr[NA], axt[NA], bdr[NB], bxt[NB];
    unsigned aconst[NA], bconst[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int pr = pra + i * RPA;
        adr[i] = (S > 1 || R == 1) ? 0 : (int)divWt.div((unsigned)pr);
        axt[i] = pr - adr[i] * Wt;
        if ((S == 1 && adr[i] >= R) || pr >= L) axt[i] = Wt;   // beyond the slice: never valid
        aconst[i] = ((unsigned)(adr[i] * RS + XS * axt[i]) * (unsigned)ld_dy + (unsigned)coA) * 4u;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pr = prb + i * RPB;
        bdr[i] = (S > 1 || R == 1) ? 0 : (int)divWt.div((unsigned)pr);
        bxt[i] = pr - bdr[i] * Wt;
        if ((S == 1 && bdr[i] >= R) || pr >= L) bxt[i] = Wt;
        bconst[i] = ((unsigned)((bdr[i] + rB) * W + 4 * bxt[i]) * (unsigned)Cin + (unsigned)ciB) * 4u;
    }

    f32x4 ra[NA], rb0[NB], rb1[NB], rb2[NB], rb3[NB];   // one register stage: in flight for one whole K step
    int ls = sbeg;                                       // next slice to load (uniform)

    // slice -> uniform (first image row index row0 = n*H + y0, first column group xbase)
    auto slice_origin = [&](int sl, int& row0, int& xbase, int& y0) {
        const int q = (int)divS.div((unsigned)sl);     // S == 1: q = sl
        row0 = q * R;                                  // S > 1  => R == 1
        xbase = (sl - q * S) * L;
        y0 = row0 - (int)divH.div((unsigned)row0) * H;
    };
    // The fp32 MFMAs share issue bandwidth with the vector ALU (tools/micro/mfma_peak.hip), so the per-load arithmetic is
    // kept minimal: every slice-uniform address term travels in the SGPR offset of the buffer load, dead slices and the
    // absent fourth tap read through a null resource, and a load costs a compare + a select (A operand) or an add, a
    // compare and a select (V taps: the column of the tap must lie inside the frame).
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0, 0x00020000);   // every load -> 0
    int bx4[NB];
    unsigned aconstv[NA], bconstv[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) aconstv[i] = aok ? aconst[i] : OOB;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        bx4[i] = 4 * bxt[i];
        bconstv[i] = bok ? bconst[i] : OOB;
    }
    const unsigned cs = (unsigned)Cin * 4u;
    auto issue = [&]() {
        int row0, xbase, y0;
        slice_origin(ls, row0, xbase, y0);
        const bool live = ls < send;
        const int xlim = Wt - xbase, rlim = NH - row0;                       // uniform validity limits
        const __amdgpu_buffer_rsrc_t ars = live ? ar : null_rsrc;
        const __amdgpu_buffer_rsrc_t xrs = live ? xr : null_rsrc;
        const __amdgpu_buffer_rsrc_t xr4 = (live & four) ? xr : null_rsrc;
        // A operand: tile exists (and, for the dy columns, column 4*xt + acol lies inside the frame)
        const unsigned sA = (unsigned)(((row0 - rowb) * RS + XS * xbase + (from_dy ? acol : 0)) * ld_dy) * 4u;
        const int xlimA = from_dy ? min(xlim, (W - 4 * xbase - acol + 3) >> 2) : xlim;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            bool ok = axt[i] < xlimA;
            if (MULTIROW) ok = ok & (adr[i] < rlim);
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, ok ? aconstv[i] : OOB, sA, 0));
        }
        // V taps: column d_j of group xt is x column 4*xt - 1 + j
        const unsigned sB = (unsigned)((((row0 - rowb) * W + 4 * xbase) * Cin) * 4);       // tap j adds j*cs (window base: -1 row, -1 pixel)
        const int xoff = 4 * xbase - 1;
        const int xl = ((unsigned)(y0 + rB - 1) < (unsigned)H) ? xlim : 0;   // one image row per slice: the kernel row decides
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bool ok;
            if (MULTIROW) {                                                   // narrow frames only (compile-time)
                const int yy = y0 + bdr[i];
                const int y = yy - (int)divH.div((unsigned)yy) * H;
                ok = ((unsigned)(y + rB - 1) < (unsigned)H) & (bxt[i] < xlim) & (bdr[i] < rlim);
            } else {
                ok = bxt[i] < xl;
            }
            const unsigned o = bconstv[i];
            rb0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                xrs, (ok & ((unsigned)(bx4[i] + (xoff + j0)) < (unsigned)W)) ? o : OOB, sB + j0 * cs, 0));
            rb1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                xrs, (ok & ((unsigned)(bx4[i] + (xoff + j1)) < (unsigned)W)) ? o : OOB, sB + j1 * cs, 0));
            rb2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                xrs, (ok & ((unsigned)(bx4[i] + (xoff + j2)) < (unsigned)W)) ? o : OOB, sB + j2 * cs, 0));
            rb3[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                xr4, (ok & ((unsigned)(bx4[i] + (xoff + 4)) < (unsigned)W)) ? o : OOB, sB + 4 * cs, 0));
        }
        ++ls;
    };
    auto store_stage = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&dst[(pra + i * RPA) * BM + cva * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(c0f, rb0[i][j], fmaf(c1f, rb1[i][j], fmaf(c2f, rb2[i][j], rb3[i][j])));
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

    // A wave's two 32-wide MFMA tiles along a dimension take INTERLEAVED rows/columns (tile t owns 2*lane + t): both
    // operands of a lane are then adjacent in LDS and arrive with one ds_read_b64 whose 16-bit immediate offset covers the
    // whole stage — the blocked assignment (tile t owns t*32 + lane) cost a ds_read2_b32 plus a v_add per fetch, and the
    // fp32 MFMAs pay for every VALU instruction.  The epilogue maps (tile, lane) back accordingly.
    auto mma_part = [&](const float* acol_, const float* bcol_, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            float a[TM], b[TN];
            if (TM == 2) {
                const f32x2 av = *reinterpret_cast<const f32x2*>(acol_ + 2 * s * BM);
                a[0] = av[0]; a[TM - 1] = av[1];
            } else {
                a[0] = acol_[2 * s * BM];
            }
            if (TN == 2) {
                const f32x2 bv = *reinterpret_cast<const f32x2*>(bcol_ + 2 * s * BN);
                b[0] = bv[0]; b[TN - 1] = bv[1];
            } else {
#pragma unroll
                for (int t = 0; t < TN; ++t) b[t] = bcol_[2 * s * BN + t * 32];
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };

    const int nK = send - sbeg;
    issue();
    store_stage(smem);
    issue();
    __syncthreads();
    const int aco = lh * BM + wm * TM * 32 + (TM == 2 ? 2 * li : li);
    const int bco = BK * BM + lh * BN + wn * TN * 32 + (TN == 2 ? 2 * li : li);
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    // Imposed instruction order of a K step (128-row tile: 60-64 MFMAs): four phases separated by scheduling fences, each
    // with its own interleave — 16 MFMAs | the 8 LDS stores of the next slice (their transform FMAs and vmcnt waits) under
    // 16 MFMAs | the NA + 4*NB loads of the slice after that, one per MFMA | the rest.  Left alone (or with one
    // sched_group_barrier list for the whole step) hipcc stores at the top and issues every load at the bottom of the
    // step, so a slice is in flight for a barrier's length instead of a whole step.
    //   masks: 0x008 MFMA, 0x020 VMEM read, 0x200 DS write
#define CVK_FENCE() __builtin_amdgcn_sched_barrier(0)
#define CVK_WW4_STEP(cur, nxt)                                                    \
    do {                                                                          \
        if (TM * TN == 4) {                                                       \
            mma_part(cur + aco, cur + bco, 0, 4);                                 \
            CVK_FENCE();                                                          \
            store_stage(nxt);            /* slice ks+1 */                         \
            mma_part(cur + aco, cur + bco, 4, 8);                                 \
            _Pragma("unroll") for (int q_ = 0; q_ < NA + NB; ++q_) {              \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                \
                __builtin_amdgcn_sched_group_barrier(0x008, 16 / (NA + NB), 0);   \
            }                                                                     \
            CVK_FENCE();                                                          \
            issue();                     /* slice ks+2 */                         \
            mma_part(cur + aco, cur + bco, 8, 13);                                \
            _Pragma("unroll") for (int q_ = 0; q_ < NA + 4 * NB; ++q_) {          \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
            }                                                                     \
            CVK_FENCE();                                                          \
            mma_part(cur + aco, cur + bco, 13, L / 2);                            \
        } else {                         /* 64-row tile, 30-32 MFMAs: 2 | 6 stores under 6 | 18 loads under 18 | rest */ \
            mma_part(cur + aco, cur + bco, 0, 1);                                 \
            CVK_FENCE();                                                          \
            store_stage(nxt);                                                     \
            mma_part(cur + aco, cur + bco, 1, 4);                                 \
            _Pragma("unroll") for (int q_ = 0; q_ < NA + NB; ++q_) {              \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
            }                                                                     \
            CVK_FENCE();                                                          \
            issue();                                                              \
            mma_part(cur + aco, cur + bco, 4, 13);                                \
            _Pragma("unroll") for (int q_ = 0; q_ < NA + 4 * NB; ++q_) {          \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
            }                                                                     \
            CVK_FENCE();                                                          \
            mma_part(cur + aco, cur + bco, 13, L / 2);                            \
        }                                                                         \
        __syncthreads();                                                          \
        CVK_FENCE();                                                              \
    } while (0)
    int ks = 0;
    for (; ks + 2 <= nK; ks += 2) {
        CVK_WW4_STEP(buf0, buf1);
        CVK_WW4_STEP(buf1, buf0);
    }
    if (ks < nK) CVK_WW4_STEP(buf0, buf1);
#undef CVK_WW4_STEP
#undef CVK_FENCE

    float* out = slab + ((size_t)split * 6 + xi) * Cout * K3;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * TN * 32 + (TN == 2 ? 2 * li + tn : tn * 32 + li);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int im = (r & 3) + 8 * (r >> 2) + 4 * lh;           // row of the 32x32 MFMA tile
                const int row = c0 + wm * TM * 32 + (TM == 2 ? 2 * im + tm : tm * 32 + im);
                if (row < Cout && col < K3) out[(size_t)row * K3 + col] = acc[tm][tn][r];
            }
        }
}

// dw[co][r][s][ci] from the slabs: P_xi = sum over splits (fixed order), then G^T
__global__ void k_wgrad_wino4_reduce(const float* __restrict__ slab, float* __restrict__ dw, int splits, int Cout, int Cin,
                                     int Cin_pad) {
    const size_t total = (size_t)Cout * 3 * Cin;
    const size_t plane = (size_t)Cout * 3 * Cin_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t cr = i / Cin;  // co*3 + r
        const float* p = slab + cr * Cin_pad + ci;
        float P[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
#pragma unroll
            for (int x = 0; x < 6; ++x) P[x] += p[((size_t)s * 6 + x) * plane];
        }
        const float s12 = P[1] + P[2], d12 = P[2] - P[1], s34 = P[3] + P[4], d34 = P[3] - P[4];
        float* o = dw + (cr * 3) * Cin + ci;
        o[0] = 0.25f * P[0] - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
        o[Cin] = d12 * (1.f / 6.f) + d34 * (1.f / 12.f);
        o[2 * (size_t)Cin] = (s34 - s12) * (1.f / 6.f) + P[5];
    }
}

struct WW4Plan { int bm, tilesM, tilesN, splits, chunk, nslices, S, R, L; };
WW4Plan plan_wgrad_wino4(int NH, int Wt, int Cin_pad, int Cout) {
    WW4Plan p;
    p.bm = Cout > 64 ? 128 : 64;
    p.tilesM = cvk_cdiv(Cout, p.bm);
    p.tilesN = cvk_cdiv(3 * Cin_pad, 128);
    auto useful = [&](int L) { return Wt >= L ? (double)Wt / ((double)cvk_cdiv(Wt, L) * L) : (double)((L / Wt) * Wt) / L; };
    p.L = useful(30) > useful(32) + 1e-9 ? 30 : 32;
    p.S = Wt >= p.L ? cvk_cdiv(Wt, p.L) : 1;               // slices per image row (wide frames)
    p.R = Wt >= p.L ? 1 : p.L / Wt;                        // whole image rows per slice (narrow frames)
    p.nslices = Wt >= p.L ? NH * p.S : cvk_cdiv(NH, p.R);
    const int units = p.tilesM * p.tilesN * 6;
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

extern "C" int cvk_wino4_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream) {
    CVK_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "cvk_wino4_weight_transform: bad arguments");
    const size_t total = (size_t)Cout * 3 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wino4_weight, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin);
    CVK_LAUNCH_RETURN("cvk_wino4_weight_transform");
}

extern "C" int cvk_wino4_weight_transform_dgrad(const float* w, float* U, int Cout, int Cin, void* stream) {
    CVK_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "cvk_wino4_weight_transform_dgrad: bad arguments");
    dim3 grid(cvk_cdiv(Cin, 32), cvk_cdiv(Cout, 32), 3);
    hipLaunchKernelGGL(k_wino4_weight_dgrad, grid, dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin);
    CVK_LAUNCH_RETURN("cvk_wino4_weight_transform_dgrad");
}

// How a layer is cut into workgroups: number of six-index blocks (the rest are single-index) and the K split of those.
struct W4Plan { int tilesN, tiles, nfull, ksplit; };
static W4Plan plan_wino4(int N, int H, int W, int Cin, int ldy) {
    W4Plan p;
    const int Mt = N * H * ((W + 3) / 4);
    p.tilesN = ldy > 64 ? cvk_cdiv(ldy, 128) : 1;
    p.tiles = cvk_cdiv(Mt, 128) * p.tilesN;
    // Tiles are dealt to the 256 CUs in whole rounds as six-index blocks; the remainder is cut into single-index blocks.
    // Everything is cut that way for deep layers with few tiles but long K, and for <= 64 output columns with Cin >= 128:
    // there a block's input footprint per index (3 rows x 512 pixels x Cin) is re-read six times and the 64 blocks of an XCD
    // overflow its 4 MB L2 (PMC: 3.6 GB fetched per launch for a 0.7 GB input) — the six single-index blocks of a tile are
    // neighbours on one XCD and run at the same time, so five of the six reads hit L2 (-17 % time on the 128->64 layers).
    p.nfull = p.tiles / 256 * 256;
    if ((p.tiles < 1024 && Cin >= 256) || (ldy <= 64 && Cin >= 128)) p.nfull = 0;
    // Few single-index blocks (< 4 waves of the 512 resident workgroups): also split the K loop 2 or 3 ways when that
    // fills the last wave better (the partial planes are small for such layers).  nK / ksplit must stay even.
    p.ksplit = 1;
    const long blocks = (long)p.tiles * 6;
    if (p.nfull == 0 && blocks < 2048 && Cin >= 256) {
        const int nK = 3 * Cin / BK;
        double best = (double)blocks / (512.0 * cvk_cdiv(blocks, 512));
        for (int f = 2; f <= 3; ++f) {
            if (nK % (2 * f)) continue;
            const double fill = (double)(blocks * f) / (512.0 * cvk_cdiv(blocks * f, 512));
            if (fill > best + 0.08) { best = fill; p.ksplit = f; }
        }
    }
    return p;
}

extern "C" int cvk_conv3x3_wino4_ksplit(int N, int H, int W, int Cin, int Cout_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout_ld <= 0) return 0;
    return plan_wino4(N, H, W, Cin, Cout_ld).ksplit;
}

extern "C" size_t cvk_conv3x3_wino4_workspace_bytes(int N, int H, int W, int Cin, int Cout_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout_ld <= 0) return 0;
    return (size_t)6 * plan_wino4(N, H, W, Cin, Cout_ld).ksplit * N * H * ((W + 3) / 4) * Cout_ld * sizeof(float);
}

extern "C" int cvk_conv3x3_wino4_gemm(const float* x, const float* U, float* Mo, int N, int H, int W, int Cin, int Cout,
                                      int ldm, void* stream) {
    CVK_CHECK_ARG(x && U && Mo, "cvk_conv3x3_wino4_gemm: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldm >= Cout && ldm % 4 == 0, "cvk_conv3x3_wino4_gemm: bad shape");
    CVK_CHECK_ARG(Cin > 0 && Cin % 64 == 0, "cvk_conv3x3_wino4_gemm: Cin=%d must be a multiple of 64 (use cvk_conv3x3_fwd otherwise)", Cin);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(U) && cvk_aligned16(Mo), "cvk_conv3x3_wino4_gemm: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wino4_gemm: tensor too large for 32-bit pixel indices");
    CVK_CHECK_ARG((long)(2 * W + 520) * Cin * 4 < (1L << 31) && (long)6 * Cout * 3 * Cin * 4 < (1L << 31), "cvk_conv3x3_wino4_gemm: a tile's input window or the weight tensor exceeds the 2 GiB buffer-addressing limit");
    const int Wt = (W + 3) / 4, Mt = N * H * Wt, Mpix = N * H * W, ldy = ldm;
    hipStream_t s = (hipStream_t)stream;
    const W4Plan p = plan_wino4(N, H, W, Cin, ldy);
    const dim3 grid(p.nfull + 6 * p.ksplit * (p.tiles - p.nfull));
    if (ldy > 64)
        hipLaunchKernelGGL((k_conv3x3_wino4<128, 128, 2, 2>), grid, dim3(256), 0, s, x, U, Mo, Mt, H, W, Wt, Cin, Cout, ldy, p.tilesN, Mpix, p.nfull, p.ksplit);
    else if (ldy > 32)
        hipLaunchKernelGGL((k_conv3x3_wino4<128, 64, 2, 2>), grid, dim3(256), 0, s, x, U, Mo, Mt, H, W, Wt, Cin, Cout, ldy, p.tilesN, Mpix, p.nfull, p.ksplit);
    else   // narrow heads: 32-column tiles, four 32x32 wave tiles stacked in M
        hipLaunchKernelGGL((k_conv3x3_wino4<128, 32, 4, 1>), grid, dim3(256), 0, s, x, U, Mo, Mt, H, W, Wt, Cin, Cout, ldy, p.tilesN, Mpix, p.nfull, p.ksplit);
    CVK_LAUNCH_RETURN("cvk_conv3x3_wino4_gemm");
}

extern "C" int cvk_wino4_output(const float* Mo, const float* bias, float* y, float* stats, int N, int H, int W, int Cout,
                                int ldy, int ksplit, void* stream) {
    CVK_CHECK_ARG(Mo && y, "cvk_wino4_output: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout && ldy % 4 == 0 && ksplit >= 1 && ksplit <= 3, "cvk_wino4_output: bad shape");
    CVK_CHECK_ARG(cvk_aligned16(Mo) && cvk_aligned16(y), "cvk_wino4_output: pointers must be 16-byte aligned");
    const int Wt = (W + 3) / 4, Mt = N * H * Wt, Mpix = N * H * W;
    const int P = cvk_cdiv(Mpix, CVK_STAT_ROWS);
    // one block = one 64-pixel granule x `chunk` channels; small layers (few granules) get 256- or 64-channel blocks so
    // that the grid still covers the chip (the 22x30 level has only 83 granules)
    int chunk = 1024;
    while (chunk > 64 && (long)P * cvk_cdiv(ldy, chunk) < 1024) chunk /= 4;
    dim3 grid(P, cvk_cdiv(ldy, chunk));
    hipStream_t s = (hipStream_t)stream;
    if (stats)
        hipLaunchKernelGGL(k_wino4_output<true>, grid, dim3(256), 0, s, Mo, ldy, Mt, bias, y, ldy, stats, P, Mpix, H, W, Wt, ldy, Cout, ksplit, chunk);
    else
        hipLaunchKernelGGL(k_wino4_output<false>, grid, dim3(256), 0, s, Mo, ldy, Mt, bias, y, ldy, stats, P, Mpix, H, W, Wt, ldy, Cout, ksplit, chunk);
    CVK_LAUNCH_RETURN("cvk_wino4_output");
}

extern "C" size_t cvk_conv3x3_wgrad_wino4_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout, int ld_dy) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_pad <= 0 || Cout <= 0 || ld_dy < Cout) return 0;
    const int Wt = (W + 3) / 4;
    const WW4Plan p = plan_wgrad_wino4(N * H, Wt, Cin_pad, Cout);
    return ((size_t)4 * N * H * Wt * ld_dy + (size_t)p.splits * 6 * Cout * 3 * Cin_pad) * sizeof(float);
}

extern "C" int cvk_conv3x3_wgrad_wino4(const float* x, const float* dy, const float* E_pre, float* dw, int N, int H, int W, int Cin,
                                       int Cin_pad, int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dy && dw && workspace, "cvk_conv3x3_wgrad_wino4: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin, "cvk_conv3x3_wgrad_wino4: bad shape");
    CVK_CHECK_ARG(Cin_pad % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= Cout, "cvk_conv3x3_wgrad_wino4: Cin_pad and ld_dy must be multiples of 4, ld_dy >= Cout");
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(dy) && cvk_aligned16(workspace), "cvk_conv3x3_wgrad_wino4: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wgrad_wino4: tensor too large for 32-bit pixel indices");
    const int Wt = (W + 3) / 4, Mpix = N * H * W, K3 = 3 * Cin_pad;
    CVK_CHECK_ARG((long)N * H * H < (1L << 32) && (long)N * H * Wt < (1L << 31), "cvk_conv3x3_wgrad_wino4: frame too large for the multiply-high coordinate split");
    const WW4Plan p = plan_wgrad_wino4(N * H, Wt, Cin_pad, Cout);
    {   // x, dy and E may exceed 2 GiB: a workgroup addresses only the image rows of its own slice range (window_rsrc)
        const long rows = (p.S > 1 ? p.chunk / p.S + 2 : (long)p.chunk * p.R) + 4;
        CVK_CHECK_ARG(rows * W * (Cin_pad > ld_dy ? Cin_pad : ld_dy) * 4 < (1L << 31), "cvk_conv3x3_wgrad_wino4: one slice range exceeds the 2 GiB buffer-addressing limit");
    }
    const size_t e_floats = (size_t)4 * N * H * Wt * ld_dy;
    const size_t need = (e_floats + (size_t)p.splits * 6 * Cout * K3) * sizeof(float);
    if (workspace_bytes < need) {
        cvk_set_error("cvk_conv3x3_wgrad_wino4: workspace %zu < %zu bytes", workspace_bytes, need);
        return CVK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* E = (float*)workspace;
    float* slab = E + e_floats;
    if (E_pre != nullptr) {        // planes E1..E4 already written by cvk_bn_bwd_dx_e
        CVK_CHECK_ARG(cvk_aligned16(E_pre), "cvk_conv3x3_wgrad_wino4: E_pre must be 16-byte aligned");
    } else {
        const size_t total = (size_t)N * H * Wt * (ld_dy / 4);
        const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(k_wino4_dy_transform, dim3(blocks), dim3(256), 0, s, dy, ld_dy, E, N * H, W, Wt);
    }
    const float* Euse = E_pre != nullptr ? E_pre : E;
    dim3 grid(p.tilesM * p.tilesN * p.splits * 6);
#define CVK_WW4_LAUNCH(BM_, MR_, L_)                                                                                                       \
    hipLaunchKernelGGL((k_wgrad_wino4<BM_, 128, 2, 2, MR_, L_>), grid, dim3(256), 0, s, x, dy, Euse, slab, N * H, H, W, Wt, Cin_pad, Cout, ld_dy, \
                       K3, p.chunk, p.tilesN, Mpix, p.tilesM * p.tilesN, p.nslices, p.S, p.R)
#define CVK_WW4_PICK(BM_)                                                                         \
    do {                                                                                          \
        if (p.L == 30) { if (p.R > 1) CVK_WW4_LAUNCH(BM_, true, 30); else CVK_WW4_LAUNCH(BM_, false, 30); } \
        else { if (p.R > 1) CVK_WW4_LAUNCH(BM_, true, 32); else CVK_WW4_LAUNCH(BM_, false, 32); }           \
    } while (0)
    if (p.bm == 128) CVK_WW4_PICK(128); else CVK_WW4_PICK(64);
#undef CVK_WW4_PICK
#undef CVK_WW4_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cvk_set_error("cvk_conv3x3_wgrad_wino4: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    const size_t total = (size_t)Cout * 3 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wgrad_wino4_reduce, dim3(blocks), dim3(256), 0, s, slab, dw, p.splits, Cout, Cin, Cin_pad);
    CVK_LAUNCH_RETURN("cvk_conv3x3_wgrad_wino4");
}
