// EXPERIMENT, NOT BUILT (round 3): kept for the record of DESIGN.md "Tried and rejected in round 3".  Version 1 of this file
// (one kernel row per workgroup, 16-row slices, 132 registers) was bit-for-bit deterministic and matched fp64 to 5e-6, but ran
// 0.94-0.80x the speed of csrc/wino4.hip's k_wgrad_wino4 (tools/bench_conv.py: 64->64 @360x480 932 vs 879 us incl. its dy
// pass, 128->128 @180x240 704 vs 564 us): 3 k-cycles of MFMAs per 12-wave barrier are too few to amortise the barrier, the
// exposed first fragment read and the staging wave's ~100 vector instructions.  This version (all three kernel rows per
// workgroup, 8-row slices: 1.5x the MFMAs per barrier) needs 96 accumulator + ~85 other registers per wave at three waves
// per SIMD (170 available): hipcc spills (ScratchSize 392 B/lane) and it runs 5-8x slower.  A variant that fits needs two
// waves per SIMD with 3 x 3 x 2 x 2 blocks per wave, i.e. b64 A-fragment reads of LDS-DMA'd rows (2-way bank conflicts) —
// not done.
// wgrad4f.hip — FUSED transposed Winograd F(4,3) weight-gradient of the 3x3 convolution for the 64/128-channel levels
// (reference: the weight gradient of nn.Conv2d(cin,cout,3,padding=1), models/unet.py:11, backward of train.py:131).
//
//   dW[co][r][0..2][ci] = G^T [ sum over groups of four columns  (A dy)_xi (.) (B^T d)_xi ],  xi = 0..5   (as csrc/wino4.hip)
//   E = A dy:  E0 = dy0, E1 = dy0+dy1+dy2+dy3, E2 = dy0-dy1+dy2-dy3, E3 = dy0+2dy1+4dy2+8dy3, E4 = dy0-2dy1+4dy2-8dy3, E5 = dy3
//   V = B^T d as in the forward kernels, d_j = x[y + r - 1][4t - 1 + j]
// i.e. per transform index one GEMM  P_xi[Cout][3*Cin] = E_xi^T V_xi  whose depth is the tile-row index t (a tile row = four
// output columns of one image row): 9*M*Cin*Cout executed FLOPs instead of 18.
//
// wino4.hip's k_wgrad_wino4 gives every transform index its own workgroup: each re-loads the pixels (four loads + three FMAs per
// V element, ~110 vector instructions per 32-64 MFMAs) — and on this part a wave's vector instructions ADD to its fp32-MFMA time
// (tools/micro/mfma_xwave.hip), hence 0.57-0.62 of peak.  Here ONE workgroup computes all six indices of a 64 x 64 (co, ci)
// block, for the three kernel rows:
//   * a pixel is loaded once per depth slice and gives all six V_xi (6 loads + 12 vector ops per six elements), by ONE of the
//     three waves of a SIMD per slice (the staging role rotates), under the MFMAs of the other two;
//   * E arrives by LDS-DMA from six planes E0..E5 that the BatchNorm/ReLU-backward pass writes on its way (cvk_bn_bwd_dx_e6;
//     cvk_wino4f_dy_transform for callers that bring their own dy): no vector work at all;
//   * both operands lie in LDS as they lie in memory ([depth][channel]): v_mfma_f32_16x16x4_f32 with INTERLEAVED rows/columns —
//     a lane's A values for the four 16-row blocks are four consecutive channels (one ds_read_b128), its B values for two
//     column blocks one ds_read_b64 — 2 LDS reads per 8 MFMAs, results leave as 8-byte channel pairs.
// Workgroup = 12 waves (three per SIMD): wave = (transform index, column half), 3 kernel rows x 8 accumulator blocks each; LDS:
// two stages of (E: 6 planes + V: 3 x 6 planes) x 8 depth rows x 256 B = 96 KiB.  Grid: depth ranges x (Cin/64) x (Cout/64); partial sums per depth range in slabs
// [range][xi][Cout][3*Cin] that wino4.hip's k_wgrad_wino4_reduce sums in a fixed order and transforms with G^T (deterministic).
#include "conv_tile.h"
#include "lds_dma.h"
#include <utility>

namespace {

template <int... Ks, class F>
__device__ __forceinline__ void g_static_for(std::integer_sequence<int, Ks...>, F&& f) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

constexpr int G_BT = 8;                          // depth rows (tile rows) per slice
constexpr int G_PLANE = G_BT * 256;              // one transform index of one operand and kernel row: 8 rows x 64 channels x 4 B
constexpr int G_EOPER = 6 * G_PLANE;             // E: 12 KiB
constexpr int G_VOPER = 3 * 6 * G_PLANE;         // V: three kernel rows, 36 KiB
constexpr int G_STAGE = G_EOPER + G_VOPER;       // 48 KiB
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void g_dma16(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// E6[xi][t][c] from dy (for callers that do not get the planes from cvk_bn_bwd_dx_e6); rows t in [Mt, Mtp) are zero
__global__ void k_wino4f_dy_transform(const float* __restrict__ DY, int ld, float* __restrict__ E, int ld_e, int NH, int W, int Wt, int Mtp, int C) {
    const int cvn = C / 4;
    const size_t total = (size_t)Mtp * cvn;
    const size_t plane = (size_t)Mtp * ld_e;
    const size_t Mt = (size_t)NH * Wt;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        const size_t t = i / cvn;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 d0 = z, d1 = z, d2 = z, d3 = z;
        if (t < Mt) {
            const int xt = (int)(t % Wt);
            const size_t row = t / Wt;
            const float* p = DY + (row * W + 4 * (size_t)xt) * ld + cv * 4;
            d0 = *reinterpret_cast<const f32x4*>(p);
            if (4 * xt + 1 < W) d1 = *reinterpret_cast<const f32x4*>(p + ld);
            if (4 * xt + 2 < W) d2 = *reinterpret_cast<const f32x4*>(p + 2 * ld);
            if (4 * xt + 3 < W) d3 = *reinterpret_cast<const f32x4*>(p + 3 * ld);
        }
        float* o = E + t * ld_e + cv * 4;
        const f32x4 a = d0 + d2, b = d1 + d3, c = d0 + 4.f * d2, d = 2.f * d1 + 8.f * d3;
        *reinterpret_cast<f32x4*>(o) = d0;
        *reinterpret_cast<f32x4*>(o + plane) = a + b;
        *reinterpret_cast<f32x4*>(o + 2 * plane) = a - b;
        *reinterpret_cast<f32x4*>(o + 3 * plane) = c + d;
        *reinterpret_cast<f32x4*>(o + 4 * plane) = c - d;
        *reinterpret_cast<f32x4*>(o + 5 * plane) = d3;
    }
}

__global__ __launch_bounds__(768, 3) void k_wgrad_wino4f(
    const float* __restrict__ X, const float* __restrict__ E6, float* __restrict__ slab, int Mt, int Mtp, int H, int W, int Wt,
    int Cin_ld, int Cout, int ld_e, int Mpix, int chunk, int nci, int nco) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * G_STAGE];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave >> 1, h = wave & 1;                 // transform index, column half of the 64-column block
    const int lj = lane & 15, kq = lane >> 4;               // row / column inside a 16 x 16 MFMA block, depth quarter

    // unit = (ci block, co block), all three kernel rows; the units of one depth range are neighbours (same pixels -> one L2)
    const int units = nci * nco;
    const int id = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / units, unit = id - split * units;
    const int cit = unit % nci, cot = unit / nci;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int tb = split * chunk, te = min(Mt, tb + chunk);
    const int nsl = (te - tb + G_BT - 1) / G_BT;            // depth slices of this workgroup (>= 1 by construction)
    const int K3 = 3 * Cin_ld;

    const FastDiv divWt((unsigned)Wt), divH((unsigned)H);
    // x window: one image row + one pixel before the first pixel of the range (as csrc/wino4f.hip); the kernel row and the
    // channel block travel in the scalar offset
    const int q0 = (int)divWt.div((unsigned)tb);
    const int first = q0 * W + 4 * (tb - q0 * Wt);
    const long xfirst = ((long)first - (W + 1)) * Cin_ld;
    const size_t xbytes = (size_t)((long)Mpix * Cin_ld - xfirst) * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(X + xfirst), 0, (int)(xbytes < 0x7FFFFFFFu ? xbytes : 0x7FFFFFFFu), 0x00020000);
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0, 0x00020000);

    // ---- V staging: six of the twelve waves stage a slice (set = slice parity; waves 0-5 / 6-11), one item per lane: kernel row
    // sidx / 2, depth row (sidx % 2) * 4 + lane / 16, 16-byte channel chunk lane % 16; its pixels are loaded two steps ahead ----
    const int sset = wave / 6, sidx = wave % 6;
    const int st_r = sidx >> 1, st_row = (sidx & 1) * 4 + (lane >> 4), st_chunk = lane & 15;
    const unsigned xso = (unsigned)((st_r * W * Cin_ld + ci0) * 4);
    f32x4 d[6];
    auto load_slice = [&](int g) {                      // issue the six pixel loads of this lane's item of slice g
        const int t = tb + g * G_BT + st_row;
        unsigned fl = 0, base = 0;
        if (g < nsl && t < te) {
            const int q = (int)divWt.div((unsigned)t), xt = t - q * Wt;
            const int y = q - (int)divH.div((unsigned)q) * H;
            const bool rowok = (unsigned)(y + st_r - 1) < (unsigned)H;
            fl = rowok ? (2u | (xt > 0 ? 1u : 0u)) : 0u;                     // bit 0: column d0, bit 1: d1
#pragma unroll
            for (int j = 2; j < 6; ++j) fl |= (rowok && 4 * xt + j - 1 < W) ? (1u << j) : 0u;
            base = (unsigned)(q * W + 4 * xt - first) * (unsigned)Cin_ld * 4u + (unsigned)st_chunk * 16u;
        }
        const __amdgpu_buffer_rsrc_t xs = g < nsl ? xr : null_rsrc;
        const unsigned cs = (unsigned)Cin_ld * 4u;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, oob_unless((fl >> j) & 1u, base + j * cs), xso, 0));
    };
    // V = B^T d for all six indices, into the V planes of `stage`: kernel row st_r, row st_row, chunk st_chunk; the two 8-byte
    // halves of a chunk are swapped in odd depth rows (the B fragment reads of depth rows t, t+1 then use disjoint banks)
    auto store_half = [&](char* stage, int half) {      // half 0: V0..V2, half 1: V3..V5 (12 live registers at a time)
        f32x4 v[3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
            const int cc = (st_row & 1) ? (c ^ 2) : c;
            if (half == 0) {
                const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
                v[0][cc] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                v[1][cc] = a + b;
                v[2][cc] = a - b;
            } else {
                const float e = d4 - d2, f = d3 - d1;
                v[0][cc] = fmaf(2.f, f, e);
                v[1][cc] = fmaf(-2.f, f, e);
                v[2][cc] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
            }
        }
        char* const p = stage + G_EOPER + st_r * (6 * G_PLANE) + st_row * 256 + st_chunk * 16 + half * 3 * G_PLANE;
#pragma unroll
        for (int x = 0; x < 3; ++x) *reinterpret_cast<f32x4*>(p + x * G_PLANE) = v[x];
    };
    auto store_slice = [&](char* stage) {
        store_half(stage, 0);
        store_half(stage, 1);
    };
    // ---- E by LDS-DMA: 12 pieces of 1 KiB per slice (piece = plane wave / 2, depth rows (wave % 2) * 4 .. +3), one per wave ----
    const unsigned evoff = (unsigned)(((lane >> 4) * ld_e + (lane & 15) * 4) * 4);
    auto dma_E = [&](int g, unsigned stage_addr) {
        const int pl = wave >> 1, tr = (wave & 1) * 4;
        const int gg = min(g, nsl - 1);                   // past the last slice: copied again, never read
        const float* src = E6 + ((size_t)pl * Mtp + tb + gg * G_BT + tr) * ld_e + co0;
        g_dma16(src, evoff, stage_addr + pl * G_PLANE + tr * 256);
    };

    f32x4 acc[24];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[r * 8 + b * 2 + c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: depth row 4*ks + kq; A: 16 bytes (four row blocks), B: 8 bytes (this wave's two column blocks)
    const int a_off = xi * G_PLANE + kq * 256 + lj * 16;
    const int b_off = G_EOPER + xi * G_PLANE + kq * 256 + lj * 16 + 8 * (h ^ (kq & 1));

    char* const buf0 = smem;
    char* const buf1 = smem + G_STAGE;
    // ---- prologue: slice 0 staged (by set 0), slices 1 / 2 in flight ----
    dma_E(0, smem_addr);
    if (sset == 0) {
        load_slice(0);
        store_slice(buf0);
        load_slice(2);
    } else {
        load_slice(1);
    }
    cvk_wait_vm<6>();
    cvk_lds_retire_barrier();

#define G_SB() __builtin_amdgcn_sched_barrier(0)
    // One step = 2 depth quarters x 3 kernel rows x 8 MFMAs (16 x 16 x 4) per wave; the A fragment of a quarter serves the three
    // kernel rows.  STAGE steps (this wave's set, every second step): slot 2 the transform + the six LDS stores of the next
    // slice (first use of its pixel loads, issued two steps ago), then the wave's E-DMA piece (after that wait: hipcc's vmcnt for
    // the pixels would otherwise cover the fresh DMA too), then the pixel loads of the slice two steps on.
    auto step = [&](auto stage_, const char* cur, char* nxt, unsigned nxt_addr, int g) {
        constexpr bool STG = decltype(stage_)::value;
        f32x4 fa;
        f32x2 fb[3];
        g_static_for(std::make_integer_sequence<int, 48>{}, [&](auto kc_) {
            constexpr int k_ = decltype(kc_)::value;
            constexpr int ks = k_ / 24, r = (k_ / 8) % 3, b = (k_ >> 1) & 3, c = k_ & 1;
            if constexpr (k_ % 24 == 0) {          // fragments of this depth quarter (three waves per SIMD cover the latency)
                fa = *reinterpret_cast<const f32x4*>(cur + a_off + ks * 1024);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) fb[rr] = *reinterpret_cast<const f32x2*>(cur + b_off + rr * (6 * G_PLANE) + ks * 1024);
                G_SB();
            }
            acc[r * 8 + b * 2 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[b], fb[r][c], acc[r * 8 + b * 2 + c], 0, 0, 0);
            if constexpr (STG) {
                if constexpr (k_ == 2) store_half(nxt, 0);
                if constexpr (k_ == 4) store_half(nxt, 1);
                if constexpr (k_ == 6) dma_E(g + 1, nxt_addr);
                if constexpr (k_ == 7) load_slice(g + 3);
            } else {
                if constexpr (k_ == 2) dma_E(g + 1, nxt_addr);
            }
            G_SB();
        });
        if constexpr (STG) cvk_wait_vm<6>(); else cvk_wait_vm<0>();
        cvk_lds_retire_barrier();
    };
    for (int g = 0; g < nsl; ++g) {
        const char* cur = (g & 1) ? buf1 : buf0;
        char* nxt = (g & 1) ? buf0 : buf1;
        const unsigned nxt_addr = smem_addr + ((g & 1) ? 0 : G_STAGE);
        if (((g + 1) & 1) == sset) step(std::true_type{}, cur, nxt, nxt_addr, g);
        else step(std::false_type{}, cur, nxt, nxt_addr, g);
    }
#undef G_SB
    cvk_wait_vm<0>();

    // P_xi block -> slab[split][xi][co][r * Cin_ld + ci]: row block b, register e, lane (lj, kq): co = 4 (4 kq + e) + b,
    // column blocks 2h, 2h+1 at lane lj: ci = 4 lj + 2h + {0, 1}
    float* const out = slab + ((size_t)(split * 6 + xi) * Cout + co0) * K3 + ci0 + 4 * lj + 2 * h;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 4 * (4 * kq + e) + b;
                const f32x2 v = {acc[r * 8 + b * 2][e], acc[r * 8 + b * 2 + 1][e]};
                *reinterpret_cast<f32x2*>(out + (size_t)co * K3 + r * Cin_ld) = v;
            }
}

// dw[co][r][s][ci] from the slabs: P_xi = sum over depth ranges (fixed order), then G^T  (same arithmetic as wino4.hip's reduce)
__global__ void k_wgrad_wino4f_reduce(const float* __restrict__ slab, float* __restrict__ dw, int splits, int Cout, int Cin,
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

struct G4Plan { int nci, nco, units, splits, chunk, Mtp; };
G4Plan plan_wgrad4f(int Mt, int Cin_ld, int Cout) {
    G4Plan p;
    p.nci = Cin_ld / 64;
    p.nco = Cout / 64;
    p.units = p.nci * p.nco;
    // one workgroup per CU (96 KiB of LDS): two rounds of the 256 CUs (the slabs grow with the number of depth ranges), at
    // least 16 depth slices per workgroup
    int s = cvk_cdiv(512, p.units);
    const int smax = Mt / (G_BT * 16) > 0 ? Mt / (G_BT * 16) : 1;
    if (s > smax) s = smax;
    p.chunk = cvk_cdiv(cvk_cdiv(Mt, s), G_BT) * G_BT;
    p.splits = cvk_cdiv(Mt, p.chunk);
    p.Mtp = p.splits * p.chunk;                        // rows of the E planes (zero beyond Mt)
    return p;
}

}  // namespace

extern "C" int cvk_wgrad_wino4f_e_rows(int N, int H, int W, int Cin_ld, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_ld < 64 || Cout < 64) return 0;
    return plan_wgrad4f(N * H * ((W + 3) / 4), Cin_ld, Cout).Mtp;
}

extern "C" size_t cvk_conv3x3_wgrad_wino4f_workspace_bytes(int N, int H, int W, int Cin_ld, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_ld < 64 || Cout < 64) return 0;
    const G4Plan p = plan_wgrad4f(N * H * ((W + 3) / 4), Cin_ld, Cout);
    return ((size_t)6 * p.Mtp * Cout + (size_t)p.splits * 6 * Cout * 3 * Cin_ld) * sizeof(float);
}

// E6_pre: NULL (the call builds the six planes from dy in the workspace) or the planes [6][cvk_wgrad_wino4f_e_rows][Cout]
// written by cvk_bn_bwd_dx_e6.
extern "C" int cvk_conv3x3_wgrad_wino4f(const float* x, const float* dy, const float* E6_pre, float* dw, int N, int H, int W, int Cin,
                                        int Cin_ld, int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && dw && workspace && (dy || E6_pre), "cvk_conv3x3_wgrad_wino4f: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= Cin_ld, "cvk_conv3x3_wgrad_wino4f: bad shape");
    CVK_CHECK_ARG(Cin_ld % 64 == 0 && Cout % 64 == 0 && Cout >= 64 && ld_dy >= Cout && ld_dy % 4 == 0,
                  "cvk_conv3x3_wgrad_wino4f: Cin_ld=%d and Cout=%d must be multiples of 64", Cin_ld, Cout);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(workspace) && (!dy || cvk_aligned16(dy)) && (!E6_pre || cvk_aligned16(E6_pre)),
                  "cvk_conv3x3_wgrad_wino4f: pointers must be 16-byte aligned");
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "cvk_conv3x3_wgrad_wino4f: tensor too large for 32-bit pixel indices");
    const int Wt = (W + 3) / 4, Mt = N * H * Wt, Mpix = N * H * W;
    CVK_CHECK_ARG((long)Mt * Wt < (1L << 32) && (long)N * H * H < (1L << 32), "cvk_conv3x3_wgrad_wino4f: frame too large for the multiply-high coordinate split");
    const G4Plan p = plan_wgrad4f(Mt, Cin_ld, Cout);
    CVK_CHECK_ARG(((long)p.chunk * 4 + 3L * W + 8) * Cin_ld * 4 < (1L << 31), "cvk_conv3x3_wgrad_wino4f: a depth range's input window exceeds the 2 GiB buffer-addressing limit");
    const size_t e_floats = (size_t)6 * p.Mtp * Cout;
    const size_t need = (e_floats + (size_t)p.splits * 6 * Cout * 3 * Cin_ld) * sizeof(float);
    if (workspace_bytes < need) {
        cvk_set_error("cvk_conv3x3_wgrad_wino4f: workspace %zu < %zu bytes", workspace_bytes, need);
        return CVK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* E = (float*)workspace;
    float* slab = E + e_floats;
    if (E6_pre == nullptr) {
        const size_t total = (size_t)p.Mtp * (Cout / 4);
        const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(k_wino4f_dy_transform, dim3(blocks), dim3(256), 0, s, dy, ld_dy, E, Cout, N * H, W, Wt, p.Mtp, Cout);
    }
    const float* Euse = E6_pre != nullptr ? E6_pre : E;
    hipLaunchKernelGGL(k_wgrad_wino4f, dim3(p.units * p.splits), dim3(768), 0, s, x, Euse, slab, Mt, p.Mtp, H, W, Wt, Cin_ld, Cout, Cout, Mpix,
                       p.chunk, p.nci, p.nco);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cvk_set_error("cvk_conv3x3_wgrad_wino4f: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    const size_t total = (size_t)Cout * 3 * Cin;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wgrad_wino4f_reduce, dim3(blocks), dim3(256), 0, s, slab, dw, p.splits, Cout, Cin, Cin_ld);
    CVK_LAUNCH_RETURN("cvk_conv3x3_wgrad_wino4f");
}
