// wino4f.hip — FUSED 1-D Winograd F(4,3) 3x3 convolution for the 64/128-channel full-resolution levels
// (reference operator: nn.Conv2d(cin,cout,3,padding=1), models/unet.py:11, models/segnet.py:8; its data-grad, bwd of train.py:131).
//
// Same arithmetic as wino4.hip (V = B^T d, U = G g, y = A^T m, interpolation points 0, +-1, +-2, inf; 9*M*Cin*Cout executed
// FLOPs instead of 18), but ONE workgroup computes all six transform indices of its tile and keeps the six product tiles
// M_0..M_5 in accumulator registers (6 x 32 per lane), so
//   * the output transform, the bias and the BatchNorm statistics happen in registers: the six product planes (1.5x the output,
//     written and read back by wino4.hip's k_conv3x3_wino4 + k_wino4_output) never exist — x is read once, y written once;
//   * a pixel is loaded once per K slice for ALL six V_xi (6 loads and 12 vector ops per six V elements; the per-index kernel
//     issued 20 loads and 18 FMAs for them) — round 3's microbenchmark (tools/micro/mfma_xwave.hip) shows that on this part
//     every vector-ALU instruction takes >= 2 cycles of fp32-MFMA time whichever wave issues it, so the only lever is fewer of
//     them per MFMA;
//   * the transformed filter slices arrive by LDS-DMA in the exact LDS image order (k_wino4f_weight writes them that way):
//     no registers, no ds_write, no address arithmetic.
// Tile: 128 tile rows (a tile row = four output columns of one image row) x 64 output channels; K slice = 16 input channels of
// one kernel row; LDS: two stages of 6 x (128 + 64) rows x 64 B = 144 KiB -> one workgroup (4 waves, one per SIMD) per CU,
// so a workgroup WALKS a contiguous range of tiles (persistent strip): the slice pipeline never drains between tiles and the
// epilogue's stores retire under the next tile's MFMAs.
#include "conv_tile.h"
#include "lds_dma.h"
#include "split_fmt.h"
#include <utility>

namespace {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>) — every index is a constant
// expression inside f (a `#pragma unroll` loop over 96 slots with per-slot branches was not unrolled: accumulators in scratch)
template <int... Ks, class F>
__device__ __forceinline__ void f_static_for(std::integer_sequence<int, Ks...>, F&& f) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

constexpr int F_BM = 128, F_BN = 64, F_BK = 16;
constexpr int F_ABYTES = 6 * F_BM * 64;          // six V_xi tiles, 64-byte rows
constexpr int F_BBYTES = 6 * F_BN * 64;          // six U_xi tiles
constexpr int F_STAGE = F_ABYTES + F_BBYTES;     // 73728
// per-channel constants of the epilogue (bias; BNR: scale, shift, mean, rstd of the producer's BatchNorm) live in LDS, copied once before the
// K loop: ANY vector-memory load in the epilogue — even the single conditional `bias[col]` — made hipcc's wait-count bookkeeping give up at
// the loop header and put an `s_waitcnt vmcnt(0)` into the first K step of every pair (round 6, read off the ISA of all variants: the pixel
// loads then had half a step instead of almost two to arrive).  Cout <= F_CMAX (host check).
constexpr int F_CMAX = 512;
constexpr int F_CST = 5 * F_CMAX * 4;            // bias | scale | shift | mean | rstd

// Uf[slice][n-tile][xi][64 rows][16 floats]: row = output channel (n-tile * 64 + row), the 16 floats = K elements
// slice*16 .. +15 of (G g)_xi, K = kernel row * Ck + input channel; 16-byte chunk c of row r is stored at c ^ ((r >> 2) & 3)
// (the LDS image: the sixteen rows of a ds_read_b128 phase then hit sixteen different bank groups).  Rows >= Cn are zero.
// dgrad: w is the FORWARD filter [Ck][3][3][Cn] and g_s = w[k][2-r][2-s][n] (rotated by 180 degrees, channels exchanged).
// H2 (the opt-in fp16 split-operand form, csrc/split_fmt.h): the same image with 16-bit elements — a 64-byte row holds the two fp16 terms of its 16
// K elements, 2^e(xi) * u = h1 + h2: chunks [h1 k0-7 | h1 k8-15 | h2 k0-7 | h2 k8-15], chunk c of row r at c ^ ((r >> 2) & 3) as before; e(xi) from the
// filter's largest magnitude (amax block) and the row sums of G (1-D: cvk_split_exp(amax, c_G[xi], 0)).
template <bool H2 = false>
__device__ __forceinline__ void wino4f_weight_body(const float* __restrict__ w, float* __restrict__ Uf, int Cn, int Ck, int tilesN, int dgrad,
                                                   unsigned vblock, unsigned nblocks, const unsigned* __restrict__ amaxW = nullptr,
                                                   const CvkSplitTab& tabG = CvkSplitTab{}) {
    float sc[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (H2) {
        const unsigned am = cvk_amax_read(amaxW);
#pragma unroll
        for (int x = 0; x < 6; ++x) sc[x] = cvk_pow2f(cvk_split_exp(am, cvk_split_tab_c(tabG, x), 0));
    }
    const int Nn = tilesN * F_BN;
    const size_t total = (size_t)Nn * 3 * Ck;
    for (size_t i = (size_t)vblock * blockDim.x + threadIdx.x; i < total; i += (size_t)nblocks * blockDim.x) {
        int n, r, k;
        if (dgrad) {       // n fastest: coalesced reads along the forward filter's input channels
            n = (int)(i % Nn);
            const size_t q = i / Nn;
            k = (int)(q % Ck);
            r = (int)(q / Ck);
        } else {           // k fastest
            k = (int)(i % Ck);
            const size_t q = i / Ck;
            r = (int)(q % 3);
            n = (int)(q / 3);
        }
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        if (n < Cn) {
            if (dgrad) {
                const float* g = w + ((size_t)(k * 3 + (2 - r)) * 3) * Cn + n;
                g0 = g[2 * (size_t)Cn]; g1 = g[Cn]; g2 = g[0];
            } else {
                const float* g = w + ((size_t)(n * 3 + r) * 3) * Ck + k;
                g0 = g[0]; g1 = g[Ck]; g2 = g[2 * (size_t)Ck];
            }
        }
        const int kk = r * Ck + k, slice = kk >> 4, kf = kk & 15;
        const int tn = n >> 6, nl = n & 63;
        if (H2) {
            const float u[6] = {(float)(0.25 * g0), (float)(-(g0 + g1 + g2) / 6.0), (float)(-(g0 - g1 + g2) / 6.0),
                                (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0), (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0), (float)g2};
            _Float16* const Uh = reinterpret_cast<_Float16*>(Uf);
            const int swz = (nl >> 2) & 3;
            const size_t row = (((size_t)(slice * tilesN + tn) * 6) * 64 + nl) * 32;
            const int p1 = (((kf >> 3) ^ swz) << 3) + (kf & 7), p2 = (((2 + (kf >> 3)) ^ swz) << 3) + (kf & 7);
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                const float v = u[x] * sc[x];
                const _Float16 h1 = (_Float16)v;
                const _Float16 h2 = (_Float16)(v - (float)h1);
                Uh[row + (size_t)x * (64 * 32) + p1] = h1;
                Uh[row + (size_t)x * (64 * 32) + p2] = h2;
            }
            continue;
        }
        const size_t o = (((size_t)(slice * tilesN + tn) * 6) * 64 + nl) * 16 + (((kf >> 2) ^ ((nl >> 2) & 3)) << 2) + (kf & 3);
        const size_t ps = 64 * 16;               // plane stride between transform indices
        Uf[o] = (float)(0.25 * g0);
        Uf[o + ps] = (float)(-(g0 + g1 + g2) / 6.0);
        Uf[o + 2 * ps] = (float)(-(g0 - g1 + g2) / 6.0);
        Uf[o + 3 * ps] = (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0);
        Uf[o + 4 * ps] = (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0);
        Uf[o + 5 * ps] = (float)g2;
    }
}

__global__ __launch_bounds__(256) void k_wino4f_weight(const float* __restrict__ w, float* __restrict__ Uf, int Cn, int Ck,
                                                      int tilesN, int dgrad) {
    wino4f_weight_body(w, Uf, Cn, Ck, tilesN, dgrad, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void k_wino4h_weight(const float* __restrict__ w, float* __restrict__ Uh, int Cn, int Ck, int tilesN, int dgrad,
                                                      const unsigned* __restrict__ amaxW, CvkSplitTab tabG) {
    wino4f_weight_body<true>(w, Uh, Cn, Ck, tilesN, dgrad, blockIdx.x, gridDim.x, amaxW, tabG);
}

// All fused-F(4,3) filter transforms of a step in ONE launch (16 launches of ~5 us each in a UNet step): the jobs travel by value in the
// kernel arguments, job j owns the blocks [first[j], first[j + 1]).
struct F4WJobsDev { const float* w[CVK_WT_BATCH_MAX]; float* out[CVK_WT_BATCH_MAX]; int Cn[CVK_WT_BATCH_MAX], Ck[CVK_WT_BATCH_MAX], dgrad[CVK_WT_BATCH_MAX];
                    unsigned first[CVK_WT_BATCH_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void k_wino4f_weight_batch(const F4WJobsDev jobs) {
    int j = 0;
    while (j + 1 < jobs.n && blockIdx.x >= jobs.first[j + 1]) ++j;
    wino4f_weight_body(jobs.w[j], jobs.out[j], jobs.Cn[j], jobs.Ck[j], (jobs.Cn[j] + F_BN - 1) / F_BN, jobs.dgrad[j],
                       blockIdx.x - jobs.first[j], jobs.first[j + 1] - jobs.first[j]);
}

// LDS-DMA with a scalar base: global address = sbase + voff (32-bit per-lane byte offset), LDS = m0 + 16 * lane.
__device__ __forceinline__ void f_dma16(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// ABL: ablation switches for timing experiments only (tools/bench_wino4f_ablate.py; 0 in the product): 1 no pixel loads,
// 2 no transform + LDS stores, 4 no filter DMA, 8 no epilogue (outputs not written), 16 no MFMAs
//
// 512 threads = 8 waves, two per SIMD, in a 4 x 2 grid of 32 x 32 wave tiles (one MFMA block per transform index: 6 x 16
// accumulator registers).  On this part a wave's own vector / LDS / memory instructions simply ADD to its MFMA time (fp32 MFMAs
// run on the vector lanes: tools/micro/mfma_xwave.hip, and the ablation of the 4-wave version of this kernel: the side work of a
// K step cost the same with and without MFMAs in the stream), so the staging work of one wave has to sit under the MFMAs of
// the OTHER wave on its SIMD — with one 4-wave workgroup per CU (144 KiB of LDS) there was no other wave.
//
// BNR (data-grad launches only): the tensor written is dL/d(activation) of the layer that PRODUCED this conv's input, and that
// layer's BatchNorm backward starts with two column sums over exactly these values (csrc/bn.hip k_bn_bwd<MODE 0>):
// sum g and sum g * xhat with g = dX masked by the ReLU, xhat = (yP - mean) * rstd.  The epilogue has dX in registers: it reads
// yP (the producer's conv output, same geometry and row stride as Y) and leaves per-m-tile partial sums in `stats`
// ([2][P][Cout], the layout cvk_colsum_finalize reads) — the producer's reduce pass over dX and yP is not launched at all.
struct FBnRed {
    const float* y;       // producer's conv output [N*H*W][ldy]
    const float* scale;   // gamma * rstd, beta - mean * gamma * rstd  (the forward's apply constants: the ReLU mask)
    const float* shift;
    const float* mean;
    const float* rstd;
};

// H2: the opt-in fp16 split-operand form (csrc/split_fmt.h, runner.w2d_split = 2).  Same tile walk, loader and epilogue; the staging pass splits
// V = B^T d into two fp16 terms of 2^e(xi) * V (e from the input tensor's amax block and the row sums of B^T) and writes them into the same
// 64-byte LDS rows ([h1 k0-7 | h1 k8-15 | h2 k0-7 | h2 k8-15], chunks swizzled as before), the filter image carries the same two terms, and
// a K step is 6 x 3 = 18 v_mfma_f32_32x32x16_f16 (h2.H1 + h1.H2 + h1.H1: same accumulator layout as the 48 v_mfma_f32_32x32x2f32 they
// replace, 0.375 of their matrix time, and they do not occupy the vector lanes); the epilogue undoes the two scales exactly.
struct FSplit {
    const unsigned* amaxX;      // amax block of the tensor X
    const unsigned* amaxW;      // ... of the filter
    CvkSplitTab tabB, tabG;
};
// VPL (forward launches of layers whose weight-grad runs the plane GEMM of csrc/wgradp.hip): the staging path computes exactly the rows
// V_xi = B^T d that k_wgradp_gemm reads, so the slices of the CENTRE kernel row (input row = the tile row's own image row) of n-tile 0 are
// also stored to the weight-grad's V planes — SLICE-MAJOR, plane[xi][Cin / 16][rows][16]: the 64 bytes a thread group of four holds for one
// plane row are contiguous with its neighbours' (a wave's store = 16 plane rows x 64 B = 1 KiB contiguous, whole 128-byte lines; round 3's
// row-major attempt wrote 64-byte half lines 256-512 B apart and lost the weight-grad's gain in the forward kernel).  Pad rows are
// zeroed by cvk_wgradp_zero_pads_sm beforehand.
struct FVpl {
    float* p;            // six slice-major planes
    long rows;           // cvk_wgradp_plane_rows(N, H, W)
    int Wtp;             // column groups per plane image row (ceil(W/4) rounded up to 8)
};
typedef unsigned u32x4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// CL: the epilogue's per-channel constants come from LDS (Cout <= F_CMAX; see F_CMAX) instead of global memory
template <bool STATS, int ABL = 0, bool BNR = false, bool H2 = false, bool VPL = false, bool CL = false>
__global__ __launch_bounds__(512, 2) void k_conv3x3_wino4f(
    const float* __restrict__ X, const float* __restrict__ Uf, const float* __restrict__ bias, float* __restrict__ Y,
    float* __restrict__ stats, float* __restrict__ counts, int Mt, int H, int W, int Wt, int Cin, int Cout, int ldy,
    int tilesN, int ntiles, int Mpix, int P, FBnRed bn, FSplit sp, FVpl vp) {
    static_assert(!(STATS && BNR), "forward statistics and the BatchNorm-backward sums are different launches");
    static_assert(!(VPL && (BNR || H2)), "the weight-grad's V planes are written by exact-fp32 forward launches only");
    static_assert(!(CL && H2) && !(VPL && !CL), "LDS constants: exact-fp32 variants; the plane-writing launch always has them");
    __shared__ __attribute__((aligned(1024))) char smem[2 * F_STAGE + 2048 + (CL ? F_CST : 0)];
    const unsigned smem_addr = cvk_lds_addr(smem);
    float* const red = reinterpret_cast<float*>(smem + 2 * F_STAGE);     // [2 sums][4 wave rows][64 channels]
    float* const cst = red + 512;                                         // [5][F_CMAX] per-channel constants (zero beyond Cout)
    if constexpr (CL) {
        const int c = threadIdx.x;                                        // 512 threads == F_CMAX
        const bool ok = c < Cout;
        cst[c] = (bias != nullptr && ok) ? bias[c] : 0.f;
        if (BNR) {
            cst[F_CMAX + c] = ok ? bn.scale[c] : 0.f;
            cst[2 * F_CMAX + c] = ok ? bn.shift[c] : 0.f;
            cst[3 * F_CMAX + c] = ok ? bn.mean[c] : 0.f;
            cst[4 * F_CMAX + c] = ok ? bn.rstd[c] : 0.f;
        }
    }                                                                     // visible after the prologue's barrier

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // Tiles (numbered m-tile major, n-tile fastest) are dealt to the XCDs in contiguous chunks and INTERLEAVED over the
    // workgroups of an XCD: workgroup i of the ng on XCD k walks tiles first + i, first + i + ng, ...  The workgroups of an
    // XCD advance in step, so at any time they cover a band of ~ng consecutive tiles (= image rows): the three reads of an
    // image row (as kernel row 0, 1, 2 of three neighbouring tiles) and the n-tiles of one pixel range hit the XCD's L2 —
    // with one contiguous range per workgroup every re-read came from the Infinity Cache and its latency was exposed.
    const int G = gridDim.x;
    const int nx = G < 8 ? G : 8;
    const int xcd = blockIdx.x % nx, wi = blockIdx.x / nx;
    const int ng = G / nx + (xcd < G % nx ? 1 : 0);                      // workgroups on this XCD (round-robin dispatch)
    const int xbeg = (int)((long)xcd * ntiles / nx), xend = (int)((long)(xcd + 1) * ntiles / nx);
    const int tbeg = xbeg + wi, tend = xend, tstride = ng;
    if (tbeg >= tend) return;
    const int ntw = (tend - tbeg + tstride - 1) / tstride;               // tiles of this workgroup
    const int nS = 3 * Cin / F_BK;                                       // K slices per tile
    const int total = ntw * nS;
    // H2: scale of V_xi and the exact inverse of the product's scale, per transform index (wave-uniform)
    float sA[6];
    CvkUnscale un[6];
#pragma unroll
    for (int x = 0; x < 6; ++x) { sA[x] = 1.f; un[x] = CvkUnscale{1.f, 1.f}; }
    if (H2) {
        const unsigned ax = cvk_amax_read(sp.amaxX), aw = cvk_amax_read(sp.amaxW);
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            const int ea = cvk_split_exp(ax, cvk_split_tab_c(sp.tabB, x), 0), eu = cvk_split_exp(aw, cvk_split_tab_c(sp.tabG, x), 0);
            sA[x] = cvk_pow2f(ea);
            un[x] = cvk_unscale(-(ea + eu));
        }
    }

    const FastDiv divWt((unsigned)Wt), divH((unsigned)H);
    // input window of the tile being loaded: starts one image row + one pixel before the tile's first pixel (possibly before
    // the tensor: only in-frame taps are ever dereferenced); a tap (kernel row r, column j) of a tile row at pixel p then sits
    // at the non-negative offset  ((p - first) + j) * Cin [VGPR part, range-checked]  +  r * W * Cin + channel base [SGPR
    // part].  Re-based per tile (scalar work), so tensors may be arbitrarily larger than the 2 GiB one resource can address.
    __amdgpu_buffer_rsrc_t xr;
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0, 0x00020000);

    // staging item of this thread: row tid >> 2 of the tile, 16-byte channel chunk tid & 3
    const int srow = tid >> 2, schunk = tid & 3;
    unsigned ibase = 0, iflag = 0;      // byte offset of (pixel d1 - first) * Cin * 4 + chunk * 16; bits 0..2 kernel rows, 3..8 columns d0..d5
    unsigned aoff[6];
    constexpr unsigned VPO_NONE = 0xFFFFFFF0u;
    unsigned vpo = VPO_NONE;            // VPL: byte offset of this thread's 16 bytes in a (xi, slice) sub-plane (out of range for rows beyond the tensor)
    auto set_tile = [&](int tile) {
        const int m0 = (tile / tilesN) * F_BM;
        const int q0 = (int)divWt.div((unsigned)m0);
        const int first = q0 * W + 4 * (m0 - q0 * Wt);
        const long xfirst = ((long)first - (W + 1)) * Cin;
        const size_t xbytes = (size_t)((long)Mpix * Cin - xfirst) * 4;
        xr = __builtin_amdgcn_make_buffer_rsrc((void*)(X + xfirst), 0, (int)(xbytes < 0x7FFFFFFFu ? xbytes : 0x7FFFFFFFu), 0x00020000);
        const int t = m0 + srow;
        unsigned fl = 0, base = 0;
        if (t < Mt) {
            const int q = (int)divWt.div((unsigned)t), xt = t - q * Wt;
            const int nimg = (int)divH.div((unsigned)q);
            const int y = q - nimg * H;
            if (VPL) vpo = ((unsigned)(vp.Wtp + (q + 2 * nimg + 1) * vp.Wtp + xt) << 6) + (unsigned)schunk * 16u;
            fl = 2u | (y > 0 ? 1u : 0u) | (y + 1 < H ? 4u : 0u);
            fl |= (xt > 0 ? 8u : 0u) | 16u;
#pragma unroll
            for (int j = 2; j < 6; ++j) fl |= (4 * xt + j - 1 < W) ? (8u << j) : 0u;
            base = (unsigned)(q * W + 4 * xt - first) * (unsigned)Cin * 4u + (unsigned)schunk * 16u;
        }
        else if (VPL) vpo = VPO_NONE;
        ibase = base;
        iflag = fl;
    };
    auto regroup = [&](int r) {                          // kernel row changed: validity of the six taps
        const unsigned cs = (unsigned)Cin * 4u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const unsigned need = (1u << r) | (8u << j);
            aoff[j] = oob_unless((iflag & need) == need, ibase + j * cs);
        }
    };

    // ---- loader state (uniform): next slice whose pixels are issued into the register stage ----
    int ltile = tbeg, lr = 0, lcib = 0, lleft = total;
    f32x4 d[2][6];        // two register stages: a slice's pixels are in flight for almost two K steps
    auto load_A = [&](int set, int j) {
        const __amdgpu_buffer_rsrc_t xs = (lleft > 0 && !(ABL & 1)) ? xr : null_rsrc;   // past the last slice: zeros, no memory access
        const unsigned so = (unsigned)((lr * W * Cin + lcib) * 4);
        d[set][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, aoff[j], so, 0));
    };
    auto advance_A = [&]() {
        --lleft;
        lcib += F_BK;
        const int w1 = lcib >= Cin;
        lcib = w1 ? 0 : lcib;
        lr += w1;
        const int w2 = lr == 3;
        lr = w2 ? 0 : lr;
        ltile += w2 ? tstride : 0;
    };
    auto issue_A = [&](int set) {
        if (lcib == 0) {       // the slice opens a new kernel row (or tile)
            if (lr == 0 && ltile < tend) set_tile(ltile);
            regroup(lr);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) load_A(set, j);
        advance_A();
    };
    // pieces of the staging work, placed one per MFMA slot by F_STEP: xform(c) computes V = B^T d of channel component c for all
    // six transform indices (12 vector ops per six elements), write_A(stage, x) stores V_x, load_A(j) refills a register
    f32x4 tv[6];
    // V = B^T d of this thread's four channels for all six transform indices, on channel PAIRS (v_pk_fma_f32 / v_pk_add_f32: 12 vector
    // instructions per pair — the vector instructions of a K step add to its matrix time one for one, tools/bench_wino4h_ablate.sh; until
    // round 5 one component per slot, 12 scalar instructions each)
    auto xform2 = [&](int set, int p) {
        if (ABL & 2) return;
        auto pr = [&](int j) { return f32x2v{d[set][j][2 * p], d[set][j][2 * p + 1]}; };
        const f32x2v d0 = pr(0), d1 = pr(1), d2 = pr(2), d3 = pr(3), d4 = pr(4), d5 = pr(5);
        // the round-3 fmaf chain, two channels per instruction, bitwise the same values.  Written as inline asm: hipcc turns half of the vector
        // expression back into scalar instructions (a packed fma survives only with an inline constant; a vector subtraction became two
        // scalar adds): 12 packed instructions per channel pair, the constants {c, c} from scalar register pairs.
        auto pkfma = [](unsigned long long cc, f32x2v u, f32x2v v) {
            f32x2v r;
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(cc), "v"(u), "v"(v));
            return r;
        };
        auto pkadd = [](f32x2v u, f32x2v v) {
            f32x2v r;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(u), "v"(v));
            return r;
        };
        constexpr unsigned long long C_M4 = 0xC0800000C0800000ull, C_M1 = 0xBF800000BF800000ull, C_P4 = 0x4080000040800000ull,
                                     C_P2 = 0x4000000040000000ull, C_M2 = 0xC0000000C0000000ull, C_M5 = 0xC0A00000C0A00000ull;
        const f32x2v a = pkfma(C_M4, d2, d4), b = pkfma(C_M4, d1, d3), e = pkfma(C_M1, d2, d4), f = pkfma(C_M1, d1, d3);
        const f32x2v t0 = pkfma(C_P4, d0, pkfma(C_M5, d2, d4)), t1 = pkadd(a, b), t2 = pkfma(C_M1, b, a), t3 = pkfma(C_P2, f, e),
                     t4 = pkfma(C_M2, f, e), t5 = pkfma(C_P4, d1, pkfma(C_M5, d3, d5));
        tv[0][2 * p] = t0[0]; tv[0][2 * p + 1] = t0[1]; tv[1][2 * p] = t1[0]; tv[1][2 * p + 1] = t1[1];
        tv[2][2 * p] = t2[0]; tv[2][2 * p + 1] = t2[1]; tv[3][2 * p] = t3[0]; tv[3][2 * p + 1] = t3[1];
        tv[4][2 * p] = t4[0]; tv[4][2 * p + 1] = t4[1]; tv[5][2 * p] = t5[0]; tv[5][2 * p + 1] = t5[1];
    };
    const int wr_off = srow * 64 + ((schunk ^ ((srow >> 2) & 3)) << 4);
    auto write_A = [&](char* stage, int x) {
        if (ABL & 2) return;
        *reinterpret_cast<f32x4*>(stage + x * (F_BM * 64) + wr_off) = tv[x];
    };
    // VPL: the slice being staged (two behind the loader: the loader's tile is still the staged slice's tile while kernel row 1 is staged,
    // Cin >= 32) — kernel row, channel offset, n-tile; and the store of V_x to sub-plane (x, channel slice)
    int st_r = 0, st_cib = F_BK, st_tn = tbeg % tilesN;
    // ONE resource over the six planes (< 4 GiB: host check); the (xi, slice) sub-plane travels in the store's scalar offset.  gfx9 range
    // check of a raw buffer: dropped if vgpr_offset >= num_records - sgpr_offset — the scalar offset COUNTS (a first version with num_records =
    // one sub-plane wrote sub-plane (0, 0) only), so num_records is the whole allocation and VPO_NONE lies beyond any of it.
    const unsigned vsub = VPL ? (unsigned)(vp.rows * 64) : 0u;            // bytes per (xi, slice) sub-plane
    const unsigned vxs = vsub * (unsigned)(Cin >> 4);                     // bytes per plane
    const unsigned long long vpa = (unsigned long long)vp.p;
    const u32x4f vr = {(unsigned)vpa, (unsigned)(vpa >> 32) & 0xFFFFu, 6u * vxs, 0x00020000u};     // raw buffer: base, stride 0, num_records, dword format
    unsigned vso = 0;                                                     // sub-plane (0, staged slice)
    auto advance_staged = [&]() {
        st_cib += F_BK;
        const int w1 = st_cib >= Cin;
        st_cib = w1 ? 0 : st_cib;
        st_r += w1;
        const int w2 = st_r == 3;
        st_r = w2 ? 0 : st_r;
        st_tn = w2 ? (st_tn + tstride) % tilesN : st_tn;
    };
    auto store_V = [&](int x) {
        if (!VPL) return;
        // Inline asm on purpose: hipcc must not SEE these stores.  They sit under a uniform branch, and every such branch costs its wait-count
        // bookkeeping one position at the merge (the pixel loads look one younger per branch): with builtin stores the transform's wait for its
        // pixel loads became vmcnt(0) — a full drain every other step.  Unseen, the stores only make hipcc's counted waits conservative (loads,
        // stores and LDS-DMA retire through ONE in-order counter: a wait for "all but the N youngest" then covers a few operations more).
        // Streaming (nt): the planes are read again only by the weight-grad (A/B on one box, whole step: nt 33.38, default policy 33.57, sc0 sc1 nt
        // 33.43 ms; the six stores spread over slots 17-37 instead of beside the LDS stores: 33.44).  tv[x] is not written again before the
        // next step's transform.
asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" : : "v"(tv[x]), "v"(vpo), "s"(vr), "s"(vso + (unsigned)x * vxs) : "memory");
    };
    // H2: the two fp16 terms of 2^e * V_x for this thread's four channels: 8 bytes each into the h1 / h2 chunk of its row
    // (H2 swizzle of the A region: bit 1 from row bit 2, bit 0 from row bit 3 — the 8 rows of a half-wave's 8-byte stores then cover all 64 banks;
    // with the fp32 image's (row >> 2) & 3 rows r and r + 4 met in the same half of their bank group: PMC 25 % of LDS cycles in conflicts)
    const int swr = (((srow >> 2) & 1) << 1) | ((srow >> 3) & 1);
    const int wr_h1 = srow * 64 + (((schunk >> 1) ^ swr) << 4) + (schunk & 1) * 8;
    const int wr_h2 = srow * 64 + (((2 + (schunk >> 1)) ^ swr) << 4) + (schunk & 1) * 8;
    auto split_A = [&](char* stage, int x) {
        if (ABL & 2) return;
        // the scale as a wave-uniform 64-bit pair {s, s}: v_pk_mul_f32 takes it from scalar registers (hipcc multiplies the four channels one by one)
        const unsigned sb = __builtin_bit_cast(unsigned, sA[x]);
        const unsigned long long sc2 = ((unsigned long long)sb << 32) | sb;
        f32x2v vlo = {tv[x][0], tv[x][1]}, vhi = {tv[x][2], tv[x][3]};
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(vlo) : "v"(vlo), "s"(sc2));
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(vhi) : "v"(vhi), "s"(sc2));
        const f32x4 v = {vlo[0], vlo[1], vhi[0], vhi[1]};
        if (ABL & 32) {         // timing experiment (WRONG values): the conversions replaced by shifts — what do v_cvt_pk_f16_f32 / v_fma_mix_f32 cost?
            typedef unsigned u32x2q __attribute__((ext_vector_type(2)));
            const u32x2q a = {(__builtin_bit_cast(unsigned, v[0]) >> 16) | (__builtin_bit_cast(unsigned, v[1]) & 0xFFFF0000u),
                              (__builtin_bit_cast(unsigned, v[2]) >> 16) | (__builtin_bit_cast(unsigned, v[3]) & 0xFFFF0000u)};
            *reinterpret_cast<u32x2q*>(stage + x * (F_BM * 64) + wr_h1) = a;
            *reinterpret_cast<u32x2q*>(stage + x * (F_BM * 64) + wr_h2) = a;
            return;
        }
        const f16x4 h1 = __builtin_convertvector(v, f16x4);
        // r = v - h1 (exact) as ONE v_fma_mix_f32 per element, which reads the fp16 operand in place: hipcc's v_cvt_f32_f16 + v_sub pair was
        // the largest single item of the K step (ablation: transform + split + stores 164 of 293 us on 64 -> 64)
        typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
        const u32x2h hp = __builtin_bit_cast(u32x2h, h1);
        f32x4 r;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hp[0]), "v"(v[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hp[0]), "v"(v[1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[2]) : "v"(hp[1]), "v"(v[2]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[3]) : "v"(hp[1]), "v"(v[3]));
        const f16x4 h2 = __builtin_convertvector(r, f16x4);
        if (ABL & 64) {         // timing experiment: the arithmetic stays, the LDS stores go (values kept alive by an empty asm)
            asm volatile("" :: "v"(h1), "v"(h2));
            return;
        }
        *reinterpret_cast<f16x4*>(stage + x * (F_BM * 64) + wr_h1) = h1;
        *reinterpret_cast<f16x4*>(stage + x * (F_BM * 64) + wr_h2) = h2;
    };
    auto store_A = [&](char* stage, int set) {
        xform2(set, 0);
        xform2(set, 1);
#pragma unroll
        for (int x = 0; x < 6; ++x) { if (H2) split_A(stage, x); else write_A(stage, x); }
    };
    // ---- filter slices by LDS-DMA: slice (bslice of tile btile) -> B region of a stage; 3 pieces of 1 KiB per wave ----
    int btn = tbeg % tilesN, bslice = 0, bleft = total;       // n-tile of the tile whose filter slices are being copied
    const unsigned bvoff = (unsigned)(wave * 3 * 1024 + lane * 16);
    const char* bsrc = nullptr;
    auto dma_B_begin = [&]() {
        // past the last slice the previous one is copied again (keeps the wait counts uniform; never read)
        bsrc = reinterpret_cast<const char*>(Uf) + (size_t)(bslice * tilesN + btn) * F_BBYTES;
        if (--bleft > 0) {
            const int w = ++bslice == nS;
            bslice = w ? 0 : bslice;
            btn = w ? (btn + tstride) % tilesN : btn;
        }
    };
    auto dma_B_piece = [&](unsigned stage_addr, int q) {
        if (!(ABL & 4)) f_dma16(bsrc + q * 1024, bvoff, stage_addr + F_ABYTES + (wave * 3 + q) * 1024);
    };

    f32x16 acc[6];
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;

    const int sw = (li >> 2) & 3;
    const int a_off = (wm * 32 + li) * 64 + ((lh ^ sw) << 4);
    const int b_off = F_ABYTES + (wn * 32 + li) * 64 + ((lh ^ sw) << 4);
    // One K step = 12 substeps (transform index x, half-slice kk) of 4 MFMAs; the two fragment reads of substep s+1 are
    // issued under the MFMAs of substep s (register double buffer), see F_STEP.
    f32x4 fa[2], fb[2];
    auto load_frag = [&](const char* st, int sidx, int slot) {
        const int x = sidx >> 1, kk = sidx & 1;
        fa[slot] = *reinterpret_cast<const f32x4*>(st + ((x * (F_BM * 64) + a_off) ^ (kk << 5)));
        fb[slot] = *reinterpret_cast<const f32x4*>(st + ((x * (F_BN * 64) + b_off) ^ (kk << 5)));
    };
    auto mfma1 = [&](int k) {                              // MFMA k of the step: substep k / 4, k-pair k % 4
        if (ABL & 16) return;
        const int sidx = k >> 2, x = sidx >> 1, slot = sidx & 1, j = k & 3;
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][j], fb[slot][j], acc[x], 0, 0, 0);
    };

    // H2: per transform index the two terms of both operands (one ds_read_b128 each: the lane's row, k half lh) and three MFMAs
    // Registers: the first terms double-buffered (read three slots ahead, under the previous index' MFMAs), the second terms single: A2 is dead
    // after the index' second MFMA, B2 after its third, and each is refilled in that slot for the next index.
    f16x8 ha1[2], hb1[2], ha2, hb2;
    const int swl = (li >> 2) & 3;                                      // filter image (written by k_wino4h_weight)
    const int swa = (((li >> 2) & 1) << 1) | ((li >> 3) & 1);           // A region (written by split_A)
    const int ha_row = (wm * 32 + li) * 64, hb_row = F_ABYTES + (wn * 32 + li) * 64;
    auto load_h1 = [&](const char* st, int x, int slot) {
        ha1[slot] = *reinterpret_cast<const f16x8*>(st + x * (F_BM * 64) + ha_row + ((lh ^ swa) << 4));
        hb1[slot] = *reinterpret_cast<const f16x8*>(st + x * (F_BN * 64) + hb_row + ((lh ^ swl) << 4));
    };
    auto load_ha2 = [&](const char* st, int x) { ha2 = *reinterpret_cast<const f16x8*>(st + x * (F_BM * 64) + ha_row + (((2 + lh) ^ swa) << 4)); };
    auto load_hb2 = [&](const char* st, int x) { hb2 = *reinterpret_cast<const f16x8*>(st + x * (F_BN * 64) + hb_row + (((2 + lh) ^ swl) << 4)); };
    auto mfma_h = [&](int k) {                             // MFMA k of the step: transform index k / 3; products h1.H1, h2.H1, h1.H2
        if (ABL & 16) return;
        const int x = k / 3, p = k - 3 * x, slot = x & 1;
        if (p == 0) acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1[slot], hb1[slot], acc[x], 0, 0, 0);
        else if (p == 1) acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha2, hb1[slot], acc[x], 0, 0, 0);
        else acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1[slot], hb2, acc[x], 0, 0, 0);
    };

    // y = A^T m in registers, + bias, stores, BatchNorm statistics partial of the tile (sum, M2 about the tile mean, count).
    // Stores are range-checked buffer stores through a window that starts at the tile's first pixel (branch-free: rows
    // beyond the tensor, ragged column groups and channels >= Cout get an out-of-range offset).
    auto epilogue = [&](int tile) {
        if (ABL & 8) {
            if (tile == -1) {
                float z = 0.f;
#pragma unroll
                for (int x = 0; x < 6; ++x)
#pragma unroll
                    for (int e = 0; e < 16; ++e) z += acc[x][e];
                Y[tid] = z;
            }
            return;
        }
        const int mt = tile / tilesN, tn = tile - mt * tilesN;
        const int m0 = mt * F_BM;
        const int col = tn * F_BN + wn * 32 + li;
        const bool cok = col < Cout;
        const float bs = CL ? cst[col & (F_CMAX - 1)] : ((bias != nullptr && cok) ? bias[col] : 0.f);
        const int qb = (int)divWt.div((unsigned)m0);
        const int pixb = qb * W + 4 * (m0 - qb * Wt);               // first pixel of the tile (uniform)
        const size_t ybytes = ((size_t)Mpix - pixb) * ldy * 4;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + (size_t)pixb * ldy), 0,
                                                                            (int)(ybytes < 0x7FFFFFFFu ? ybytes : 0x7FFFFFFFu), 0x00020000);
        const unsigned ls = (unsigned)ldy * 4u;
        float s1 = 0.f, s2 = 0.f;
        __amdgpu_buffer_rsrc_t pr = null_rsrc;
        float bsc = 0.f, bsh = 0.f, bmu = 0.f;
        if (BNR) {
            pr = __builtin_amdgcn_make_buffer_rsrc((void*)(bn.y + (size_t)pixb * ldy), 0,
                                                   (int)(ybytes < 0x7FFFFFFFu ? ybytes : 0x7FFFFFFFu), 0x00020000);
            if (CL) { bsc = cst[F_CMAX + (col & (F_CMAX - 1))]; bsh = cst[2 * F_CMAX + (col & (F_CMAX - 1))]; bmu = cst[3 * F_CMAX + (col & (F_CMAX - 1))]; }
            else if (cok) { bsc = bn.scale[col]; bsh = bn.shift[col]; bmu = bn.mean[col]; }
        }
        // g = dX where the producer's ReLU passed; accumulates sum g and sum g * (yP - mean)  (rstd multiplies once, at the end)
        auto bnacc = [&](float q, float v) {
            const float g = fmaf(q, bsc, bsh) > 0.f ? v : 0.f;
            s1 += g;
            s2 = fmaf(g, q - bmu, s2);
        };
        if ((W & 3) == 0 && m0 + F_BM <= Mt && (tn + 1) * F_BN <= Cout) {
            // whole tile, no ragged column group: pixel(t) = 4 t, every output valid.  The vector ALU only does the transform,
            // the bias and the statistics (its instructions take matrix time on this part); a row's byte offset is uniform and
            // travels in the scalar offset of the store.
            const unsigned ob = (unsigned)((4 * (wm * 32 + 4 * lh)) * ldy + col) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0_ = H2 ? acc[0][e] * un[0].a * un[0].b : acc[0][e], m1 = H2 ? acc[1][e] * un[1].a * un[1].b : acc[1][e],
                            m2 = H2 ? acc[2][e] * un[2].a * un[2].b : acc[2][e], m3 = H2 ? acc[3][e] * un[3].a * un[3].b : acc[3][e],
                            m4 = H2 ? acc[4][e] * un[4].a * un[4].b : acc[4][e], m5 = H2 ? acc[5][e] * un[5].a * un[5].b : acc[5][e];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                const float y0 = m0_ + s12 + s34, y1 = fmaf(2.f, d34, d12), y2 = fmaf(4.f, s34, s12), y3 = fmaf(8.f, d34, d12) + m5;
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x][e] = 0.f;
                const unsigned so = (unsigned)(4 * ((e & 3) + 8 * (e >> 2))) * ls;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0 + bs), yr, ob, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1 + bs), yr, ob, so + ls, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y2 + bs), yr, ob, so + 2 * ls, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y3 + bs), yr, ob, so + 3 * ls, 0);
                if (STATS) {
                    s1 += (y0 + y1) + (y2 + y3);
                    s2 += fmaf(y0, y0, y1 * y1) + fmaf(y2, y2, y3 * y3);
                }
                if (BNR) {
                    // (hipcc keeps 16 of these loads in flight; asking for 32 up front changed nothing — every workgroup reaches
                    // its epilogue at about the same time and the burst is bandwidth-, not latency-bound: +10 % on the launch,
                    // against the whole reduce pass it replaces)
                    const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, ob, so, 0));
                    const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, ob, so + ls, 0));
                    const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, ob, so + 2 * ls, 0));
                    const float q3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, ob, so + 3 * ls, 0));
                    bnacc(q0, y0 + bs); bnacc(q1, y1 + bs); bnacc(q2, y2 + bs); bnacc(q3, y3 + bs);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float m0_ = H2 ? acc[0][e] * un[0].a * un[0].b : acc[0][e], m1 = H2 ? acc[1][e] * un[1].a * un[1].b : acc[1][e],
                            m2 = H2 ? acc[2][e] * un[2].a * un[2].b : acc[2][e], m3 = H2 ? acc[3][e] * un[3].a * un[3].b : acc[3][e],
                            m4 = H2 ? acc[4][e] * un[4].a * un[4].b : acc[4][e], m5 = H2 ? acc[5][e] * un[5].a * un[5].b : acc[5][e];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                const float y0 = m0_ + s12 + s34, y1 = fmaf(2.f, d34, d12), y2 = fmaf(4.f, s34, s12), y3 = fmaf(8.f, d34, d12) + m5;
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x][e] = 0.f;
                const int q = (int)divWt.div((unsigned)t), xt = t - q * Wt;
                const int nv = (t < Mt && cok) ? W - 4 * xt : 0;       // valid columns of this group
                const unsigned o = (unsigned)((q * W + 4 * xt - pixb) * ldy + col) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0 + bs), yr, oob_unless(nv > 0, o), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1 + bs), yr, oob_unless(nv > 1, o), ls, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y2 + bs), yr, oob_unless(nv > 2, o), 2 * ls, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y3 + bs), yr, oob_unless(nv > 3, o), 3 * ls, 0);
                if (STATS) {
                    const float z0 = nv > 0 ? y0 : 0.f, z1 = nv > 1 ? y1 : 0.f, z2 = nv > 2 ? y2 : 0.f, z3 = nv > 3 ? y3 : 0.f;
                    s1 += (z0 + z1) + (z2 + z3);
                    s2 += fmaf(z0, z0, z1 * z1) + fmaf(z2, z2, z3 * z3);
                }
                if (BNR) {      // an out-of-range load returns 0; the value beside it is masked by nv as well
                    const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, oob_unless(nv > 0, o), 0, 0));
                    const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, oob_unless(nv > 1, o), ls, 0));
                    const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, oob_unless(nv > 2, o), 2 * ls, 0));
                    const float q3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, oob_unless(nv > 3, o), 3 * ls, 0));
                    bnacc(q0, nv > 0 ? y0 + bs : 0.f); bnacc(q1, nv > 1 ? y1 + bs : 0.f);
                    bnacc(q2, nv > 2 ? y2 + bs : 0.f); bnacc(q3, nv > 3 ? y3 + bs : 0.f);
                }
            }
        }
        if (BNR) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                red[wm * 64 + wn * 32 + li] = s1;
                red[256 + wm * 64 + wn * 32 + li] = s2;
            }
            __syncthreads();
            if (tid < 64) {
                const int c = tn * F_BN + tid;
                if (c < Cout) {
                    const float a = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
                    const float b = (red[256 + tid] + red[320 + tid]) + (red[384 + tid] + red[448 + tid]);
                    stats[(size_t)mt * Cout + c] = a;
                    stats[(size_t)(P + mt) * Cout + c] = b * (CL ? cst[4 * F_CMAX + c] : bn.rstd[c]);
                }
            }
        }
        if (STATS) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                red[wm * 64 + wn * 32 + li] = s1;
                red[256 + wm * 64 + wn * 32 + li] = s2;
            }
            __syncthreads();
            if (tid < 64) {
                const int c = tn * F_BN + tid;
                if (c < Cout) {
                    const int mend = min(Mt, m0 + F_BM);
                    const int nlast = (int)divWt.div((unsigned)mend) - qb;   // ragged (last-of-row) column groups in the tile
                    const float cnt = (float)(4 * (mend - m0) - nlast * (4 * Wt - W));
                    const float a = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
                    const float b = (red[256 + tid] + red[320 + tid]) + (red[384 + tid] + red[448 + tid]);
                    const float m2 = b - a * a / cnt;                  // sums exclude the bias (shift invariance)
                    const float bc = CL ? cst[c] : (bias != nullptr ? bias[c] : 0.f);
                    stats[(size_t)mt * Cout + c] = a + cnt * bc;
                    stats[(size_t)(P + mt) * Cout + c] = m2 > 0.f ? m2 : 0.f;
                    if (tid == 0 && tn == 0) counts[mt] = cnt;
                }
            }
        }
    };

    // ---- prologue: slice 0 staged, slices 1 and 2 in flight ----
    char* const buf0 = smem;
    char* const buf1 = smem + F_STAGE;
    issue_A(0);                                  // slice 0 -> set 0
    issue_A(1);                                  // slice 1 -> set 1
    dma_B_begin();
#pragma unroll
    for (int q = 0; q < 3; ++q) dma_B_piece(smem_addr, q);
    store_A(buf0, 0);
    issue_A(0);                                  // slice 2 -> set 0
    cvk_wait_vm<6>();
    cvk_lds_retire_barrier();

#define F_SB() __builtin_amdgcn_sched_barrier(0)
    // Instruction order of a K step, imposed slot by slot (a scheduling fence after every MFMA: nothing moves): 48 MFMAs per wave,
    // each followed by at most one small piece of side work.  Slots 4s: the two fragment reads of substep s+1.  Slots 1,2,3,5: the
    // transform of the slice staged in this step (12 vector ops per channel component; the first waits for its pixel loads, issued
    // almost two steps earlier); 2,3,5 also one of the wave's three filter-DMA pieces each (AFTER that wait — hipcc's vmcnt for
    // the pixels would otherwise wait for the fresh DMA too — and before this step's pixel loads: vmcnt order); 6-13 (not 8, 12):
    // one LDS store of V and one pixel load (slice g + 3) into the register just freed.  Left to hipcc (with or without
    // sched_group_barrier lists) the fragment reads sit right in front of their MFMAs and the transform runs as one block with
    // the matrix pipe idle.  Tried without gain: waves 4-7 doing their side work half a step later than waves 0-3.
    auto step = [&](auto par_, const char* cur, char* nxt, unsigned nxt_addr) {
        constexpr int SET = 1 - decltype(par_)::value;      // register set holding the slice this step stages (slice g + 1)
        // VPL: does this step stage a centre-kernel-row slice of n-tile 0?  Then its six LDS stores of V are followed by six plane stores —
        // a uniform branch around each (two copies of the whole step, selected per step, made hipcc spill 134 registers)
        const bool VST = VPL && st_r == 1 && st_tn == 0;
        if (VPL) vso = (unsigned)(st_cib >> 4) * vsub;
        if (lcib == 0) {       /* the slice issued in this step (g + 3) opens a new kernel row (or tile) */
            if (lr == 0 && ltile < tend) set_tile(ltile);
            regroup(lr);
        }
        dma_B_begin();
        load_frag(cur, 0, 0);
        F_SB();
        f_static_for(std::make_integer_sequence<int, 48>{}, [&](auto kc_) {
            constexpr int k_ = decltype(kc_)::value;
            constexpr int q_ = k_;
            mfma1(k_);
            if constexpr ((k_ & 3) == 0) {
                if constexpr (k_ < 44) load_frag(cur, (k_ >> 2) + 1, ((k_ >> 2) + 1) & 1);
            } else if constexpr (q_ == 1) {
                xform2(SET, 0);                             /* waits for the six pixel loads of the slice; two channels per instruction (round 5) */
            } else if constexpr (q_ == 2) {
                xform2(SET, 1);
                dma_B_piece(nxt_addr, 0);
            } else if constexpr (q_ == 3) {
                dma_B_piece(nxt_addr, 1);
            } else if constexpr (q_ == 5) {
                dma_B_piece(nxt_addr, 2);
            } else if constexpr (q_ == 6 || q_ == 7) {
                write_A(nxt, q_ - 6);
                if (VST) store_V(q_ - 6);
                load_A(SET, q_ - 6);
            } else if constexpr (q_ >= 9 && q_ <= 11) {
                write_A(nxt, q_ - 7);
                if (VST) store_V(q_ - 7);
                load_A(SET, q_ - 7);
            } else if constexpr (q_ == 13) {
                write_A(nxt, 5);
                if (VST) store_V(5);
                load_A(SET, 5);
                advance_A();
            }
            F_SB();
        });
        // the step's DMA pieces (and, older, the previous step's pixel loads) have landed; a plane-store step leaves its six stores in
        // flight beside its six pixel loads (they retire under the next step: loads, stores and DMA share one in-order counter)
        if (VST) cvk_wait_vm<12>(); else cvk_wait_vm<6>();
        cvk_lds_retire_barrier();
        if (VPL) advance_staged();
    };
    // H2: 18 MFMA slots.  Slots 3x: the four fragment reads of transform index x + 1; 1, 2, 4, 5: the transform (the first waits for its pixel
    // loads) with the three filter-DMA pieces behind that wait; 7, 8, 10, 11, 13, 14: split + two LDS stores of one V_x and one pixel load into
    // the register just freed
    auto step_h = [&](auto par_, const char* cur, char* nxt, unsigned nxt_addr) {
        constexpr int SET = 1 - decltype(par_)::value;
        if (lcib == 0) {
            if (lr == 0 && ltile < tend) set_tile(ltile);
            regroup(lr);
        }
        dma_B_begin();
        load_h1(cur, 0, 0);
        load_ha2(cur, 0);
        load_hb2(cur, 0);
        F_SB();
        f_static_for(std::make_integer_sequence<int, 18>{}, [&](auto kc_) {
            constexpr int k_ = decltype(kc_)::value;
            mfma_h(k_);
            if constexpr (k_ < 15) {        // the next index' fragments
                if constexpr (k_ % 3 == 0) load_h1(cur, k_ / 3 + 1, (k_ / 3 + 1) & 1);
                else if constexpr (k_ % 3 == 1) load_ha2(cur, k_ / 3 + 1);
                else load_hb2(cur, k_ / 3 + 1);
            }
            if constexpr (k_ % 3 == 0) {
            } else if constexpr (k_ == 1) {
                xform2(SET, 0);                             /* waits for the six pixel loads of the slice */
            } else if constexpr (k_ == 2) {
                xform2(SET, 1);
                dma_B_piece(nxt_addr, 0);
            } else if constexpr (k_ == 4) {
                dma_B_piece(nxt_addr, 1);
            } else if constexpr (k_ == 5) {
                dma_B_piece(nxt_addr, 2);
            } else if constexpr (k_ == 7 || k_ == 10 || k_ == 13) {       /* two transform indices per slot: four independent dependency chains */
                split_A(nxt, 2 * ((k_ - 7) / 3));
                split_A(nxt, 2 * ((k_ - 7) / 3) + 1);
            } else if constexpr (k_ == 8 || k_ == 11) {
                load_A(SET, 2 * ((k_ - 8) / 3));
                load_A(SET, 2 * ((k_ - 8) / 3) + 1);
            } else if constexpr (k_ == 14) {
                load_A(SET, 4);
                load_A(SET, 5);
                advance_A();
            }
            F_SB();
        });
        cvk_wait_vm<6>();
        cvk_lds_retire_barrier();
    };
    // nS is even (Cin % 32 == 0, checked by the host): every tile starts in stage 0 and the step pair below is the only copy
    // of the K step in the code (per phase); the epilogue's global stores count in vmcnt and retire under the next tile's first step
    for (int tile = tbeg; tile < tend; tile += tstride) {
        for (int ks = 0; ks < nS; ks += 2) {
            if (H2) {
                step_h(std::integral_constant<int, 0>{}, buf0, buf1, smem_addr + F_STAGE);
                step_h(std::integral_constant<int, 1>{}, buf1, buf0, smem_addr);
            } else {
                step(std::integral_constant<int, 0>{}, buf0, buf1, smem_addr + F_STAGE);
                step(std::integral_constant<int, 1>{}, buf1, buf0, smem_addr);
            }
        }
        epilogue(tile);
    }
#undef F_SB
    cvk_wait_vm<0>();            // the redundant tail DMA lands before the workgroup's LDS is released
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------
static inline int f_tiles_n(int Cn) { return cvk_cdiv(Cn, F_BN); }

extern "C" size_t cvk_wino4f_weight_floats(int Cn, int Ck) {
    if (Cn <= 0 || Ck <= 0 || Ck % F_BK) return 0;
    return (size_t)(3 * Ck / F_BK) * f_tiles_n(Cn) * 6 * F_BN * F_BK;
}

extern "C" int cvk_wino4f_weight_transform(const float* w, float* Uf, int Cn, int Ck, int dgrad, void* stream) {
    CVK_CHECK_ARG(w && Uf && Cn > 0 && Ck > 0, "cvk_wino4f_weight_transform: bad arguments");
    CVK_CHECK_ARG(Ck % F_BK == 0, "cvk_wino4f_weight_transform: the contraction channel count %d must be a multiple of 16", Ck);
    CVK_CHECK_ARG(cvk_aligned16(Uf), "cvk_wino4f_weight_transform: Uf must be 16-byte aligned");
    const size_t total = (size_t)f_tiles_n(Cn) * F_BN * 3 * Ck;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wino4f_weight, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Uf, Cn, Ck, f_tiles_n(Cn), dgrad ? 1 : 0);
    CVK_LAUNCH_RETURN("cvk_wino4f_weight_transform");
}

extern "C" int cvk_wino4f_weight_transform_batch(const cvk_wt_job* jobs, int n, void* stream) {
    CVK_CHECK_ARG(jobs && n > 0 && n <= CVK_WT_BATCH_MAX, "cvk_wino4f_weight_transform_batch: 1..%d jobs", CVK_WT_BATCH_MAX);
    F4WJobsDev d;
    unsigned nb = 0;
    for (int i = 0; i < n; ++i) {
        const cvk_wt_job& q = jobs[i];
        CVK_CHECK_ARG(q.w && q.out && q.rows > 0 && q.cols > 0 && q.cols % F_BK == 0 && cvk_aligned16(q.out), "cvk_wino4f_weight_transform_batch: bad job %d", i);
        d.w[i] = q.w; d.out[i] = q.out; d.Cn[i] = q.rows; d.Ck[i] = q.cols; d.dgrad[i] = q.dgrad ? 1 : 0;
        const size_t total = (size_t)f_tiles_n(q.rows) * F_BN * 3 * q.cols;
        d.first[i] = nb;
        nb += (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    }
    d.first[n] = nb; d.n = n;
    hipLaunchKernelGGL(k_wino4f_weight_batch, dim3(nb), dim3(256), 0, (hipStream_t)stream, d);
    CVK_LAUNCH_RETURN("cvk_wino4f_weight_transform_batch");
}

extern "C" int cvk_wino4f_stat_partials(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return cvk_cdiv((long)N * H * ((W + 3) / 4), F_BM);
}

static int wino4f_launch(const char* who, const float* x, const float* Uf, const float* bias, float* y, float* stats, float* counts,
                         const FBnRed* bn, int N, int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream,
                         const void* amax_x = nullptr, const void* amax_w = nullptr, float* vplanes = nullptr) {
    CVK_CHECK_ARG(x && Uf && y, "%s: null pointer", who);
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout, "%s: bad shape", who);
    CVK_CHECK_ARG(Cin >= 32 && Cin % 32 == 0, "%s: Cin=%d must be a multiple of 32", who, Cin);
    CVK_CHECK_ARG((stats == nullptr) == (counts == nullptr) || bn, "%s: stats and counts go together", who);
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(Uf), "%s: x and Uf must be 16-byte aligned", who);
    CVK_CHECK_ARG((long)N * H * W < (1L << 31) - 512, "%s: tensor too large for 32-bit pixel indices", who);
    const int Wt = (W + 3) / 4, Mt = N * H * Wt, Mpix = N * H * W;
    CVK_CHECK_ARG((long)Mt * Wt < (1L << 32) && (long)N * H * H < (1L << 32), "%s: frame too large for the multiply-high coordinate split", who);
    const int tilesN = f_tiles_n(Cout), tilesM = cvk_cdiv(Mt, F_BM), ntiles = tilesM * tilesN;
    // persistent strips: one workgroup per CU (144 KiB of LDS), each walks ntiles / grid consecutive tiles
    static int cus = 0;          // queried once (a benign race: every thread stores the same value); not during a stream capture
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    // max_workgroups > 0 caps the persistent grid below one workgroup per CU: under data-parallel training RCCL's all-reduce
    // kernels hold some CUs while backward runs, and a workgroup of this kernel needs a whole CU (144 KiB of LDS, all of its
    // registers) — with a static tile assignment the workgroups that find no free CU would start only when others finish
    const int wgs = (max_workgroups > 0 && max_workgroups < cus) ? max_workgroups : cus;
    const int grid = ntiles < wgs ? ntiles : wgs;
    CVK_CHECK_ARG((F_BM * 4 + 3L * W + 8) * Cin * 4 < (1L << 31), "%s: a tile's input window exceeds the 2 GiB buffer-addressing limit", who);
    hipStream_t s = (hipStream_t)stream;
    const FBnRed none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const bool h2 = amax_x != nullptr;
    CVK_CHECK_ARG((amax_x == nullptr) == (amax_w == nullptr), "%s: the two amax blocks go together", who);
    FSplit sp = {(const unsigned*)amax_x, (const unsigned*)amax_w, CvkSplitTab{}, CvkSplitTab{}};
    if (h2) { sp.tabB = cvk_split_tab(4, CVK_SPLIT_KIND_B); sp.tabG = cvk_split_tab(4, CVK_SPLIT_KIND_G); }
    FVpl vp = {nullptr, 0, 0};
    if (vplanes) {
        CVK_CHECK_ARG(!bn && !h2 && cvk_aligned16(vplanes), "%s: V planes go with plain fp32 forward launches", who);
        vp.Wtp = (Wt + 7) / 8 * 8;
        vp.rows = (long)N * (H + 2) * vp.Wtp + 2L * vp.Wtp;
        vp.p = vplanes;
        CVK_CHECK_ARG(6L * Cin * vp.rows * 4 < (1L << 32) - 4096, "%s: V planes of %ld rows x %d channels exceed the 4 GiB addressing limit", who, vp.rows, Cin);
    }
    const bool cl = Cout <= F_CMAX;      // the epilogue's per-channel constants fit the LDS copy (every layer of these networks outside the forced test modes)
    CVK_CHECK_ARG(!vplanes || cl, "%s: the plane-writing launch needs Cout <= %d", who, F_CMAX);
#define CVK_W4F_GO(ST_, BN_, H2_, CL_, BNV_) hipLaunchKernelGGL((k_conv3x3_wino4f<ST_, 0, BN_, H2_, false, CL_>), dim3(grid), dim3(512), 0, s, x, Uf, bias, y, \
                                                                stats, counts, Mt, H, W, Wt, Cin, Cout, ldy, tilesN, ntiles, Mpix, tilesM, BNV_, sp, vp)
#define CVK_W4F_GO2(ST_, BN_, BNV_) do { if (cl) CVK_W4F_GO(ST_, BN_, false, true, BNV_); else CVK_W4F_GO(ST_, BN_, false, false, BNV_); } while (0)
#define CVK_W4F_GOV(ST_) hipLaunchKernelGGL((k_conv3x3_wino4f<ST_, 0, false, false, true, true>), dim3(grid), dim3(512), 0, s, x, Uf, bias, y, stats, counts, \
                                            Mt, H, W, Wt, Cin, Cout, ldy, tilesN, ntiles, Mpix, tilesM, none, sp, vp)
    if (vplanes)    { if (stats) CVK_W4F_GOV(true); else CVK_W4F_GOV(false); }
    else if (bn)    { if (h2) CVK_W4F_GO(false, true, true, false, *bn); else CVK_W4F_GO2(false, true, *bn); }
    else if (stats) { if (h2) CVK_W4F_GO(true, false, true, false, none); else CVK_W4F_GO2(true, false, none); }
    else            { if (h2) CVK_W4F_GO(false, false, true, false, none); else CVK_W4F_GO2(false, false, none); }
#undef CVK_W4F_GO
#undef CVK_W4F_GO2
#undef CVK_W4F_GOV
    CVK_LAUNCH_RETURN(who);
}

extern "C" int cvk_conv3x3_wino4f(const float* x, const float* Uf, const float* bias, float* y, float* stats, float* counts, int N,
                                  int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream) {
    return wino4f_launch("cvk_conv3x3_wino4f", x, Uf, bias, y, stats, counts, nullptr, N, H, W, Cin, Cout, ldy, max_workgroups, stream);
}

// The forward launch that also leaves the weight-grad's transformed input (see FVpl): Vsm = six SLICE-MAJOR planes
// [6][Cin / 16][cvk_wgradp_plane_rows(N,H,W)][16] whose pad rows cvk_wgradp_zero_pads_sm has zeroed; read by cvk_wgradp_gemm_sm.
// y, statistics and counts are bitwise those of cvk_conv3x3_wino4f.
extern "C" int cvk_conv3x3_wino4f_vplanes(const float* x, const float* Uf, const float* bias, float* y, float* stats, float* counts, float* Vsm,
                                          int N, int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream) {
    CVK_CHECK_ARG(Vsm, "cvk_conv3x3_wino4f_vplanes: null plane pointer");
    return wino4f_launch("cvk_conv3x3_wino4f_vplanes", x, Uf, bias, y, stats, counts, nullptr, N, H, W, Cin, Cout, ldy, max_workgroups, stream,
                         nullptr, nullptr, Vsm);
}

// The data-grad launch that also leaves the producer layer's BatchNorm-backward column sums (see FBnRed): `part` is
// float[2][cvk_wino4f_stat_partials(N,H,W)][Cout], read by cvk_colsum_finalize(part, partials, Cout, dbeta, dgamma).
// yP has y's geometry and row stride ldy.
extern "C" int cvk_conv3x3_wino4f_bnred(const float* x, const float* Uf, float* y, int N, int H, int W, int Cin, int Cout, int ldy,
                                        const float* yP, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, float* part, int max_workgroups, void* stream) {
    CVK_CHECK_ARG(yP && scale && shift && mean && rstd && part, "cvk_conv3x3_wino4f_bnred: null pointer");
    const FBnRed bn = {yP, scale, shift, mean, rstd};
    return wino4f_launch("cvk_conv3x3_wino4f_bnred", x, Uf, nullptr, y, part, nullptr, &bn, N, H, W, Cin, Cout, ldy, max_workgroups, stream);
}

// ---- the opt-in fp16 split-operand form (csrc/split_fmt.h; runner.w2d_split = 2): same contracts, plus the amax blocks of x and of the filter ----
// Uh: cvk_wino4f_weight_floats(Cn, Ck) * 4 bytes (the same image size: two fp16 terms instead of one fp32 value)
extern "C" int cvk_wino4h_weight_transform(const float* w, void* Uh, const void* amax_w, int Cn, int Ck, int dgrad, void* stream) {
    CVK_CHECK_ARG(w && Uh && amax_w && Cn > 0 && Ck > 0, "cvk_wino4h_weight_transform: bad arguments");
    CVK_CHECK_ARG(Ck % F_BK == 0, "cvk_wino4h_weight_transform: the contraction channel count %d must be a multiple of 16", Ck);
    CVK_CHECK_ARG(cvk_aligned16(Uh), "cvk_wino4h_weight_transform: Uh must be 16-byte aligned");
    const size_t total = (size_t)f_tiles_n(Cn) * F_BN * 3 * Ck;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_wino4h_weight, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)Uh, Cn, Ck, f_tiles_n(Cn), dgrad ? 1 : 0,
                       (const unsigned*)amax_w, cvk_split_tab(4, CVK_SPLIT_KIND_G));
    CVK_LAUNCH_RETURN("cvk_wino4h_weight_transform");
}
extern "C" int cvk_conv3x3_wino4h(const float* x, const void* Uh, const float* bias, float* y, float* stats, float* counts, const void* amax_x,
                                  const void* amax_w, int N, int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream) {
    CVK_CHECK_ARG(amax_x && amax_w, "cvk_conv3x3_wino4h: null amax block");
    return wino4f_launch("cvk_conv3x3_wino4h", x, (const float*)Uh, bias, y, stats, counts, nullptr, N, H, W, Cin, Cout, ldy, max_workgroups, stream,
                         amax_x, amax_w);
}
extern "C" int cvk_conv3x3_wino4h_bnred(const float* x, const void* Uh, float* y, const void* amax_x, const void* amax_w, int N, int H, int W, int Cin,
                                        int Cout, int ldy, const float* yP, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, float* part, int max_workgroups, void* stream) {
    CVK_CHECK_ARG(yP && scale && shift && mean && rstd && part && amax_x && amax_w, "cvk_conv3x3_wino4h_bnred: null pointer");
    const FBnRed bn = {yP, scale, shift, mean, rstd};
    return wino4f_launch("cvk_conv3x3_wino4h_bnred", x, (const float*)Uh, nullptr, y, part, nullptr, &bn, N, H, W, Cin, Cout, ldy, max_workgroups, stream,
                         amax_x, amax_w);
}

#ifdef CVK_WINO4F_ABLATE
extern "C" int cvk_conv3x3_wino4f_ablate(const float* x, const float* Uf, const float* bias, float* y, int N, int H, int W, int Cin,
                                         int Cout, int ldy, int abl, void* stream) {
    const int Wt = (W + 3) / 4, Mt = N * H * Wt, Mpix = N * H * W;
    const int tilesN = f_tiles_n(Cout), tilesM = cvk_cdiv(Mt, F_BM), ntiles = tilesM * tilesN;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipStream_t s = (hipStream_t)stream;
#define CVK_ABL(A) case A: hipLaunchKernelGGL((k_conv3x3_wino4f<false, A>), dim3(grid), dim3(512), 0, s, x, Uf, bias, y, nullptr, nullptr, Mt, H, W, Wt, Cin, Cout, ldy, tilesN, ntiles, Mpix, tilesM, FBnRed{nullptr, nullptr, nullptr, nullptr, nullptr}, FSplit{nullptr, nullptr, CvkSplitTab{}, CvkSplitTab{}}, FVpl{nullptr, 0, 0}); break;
    switch (abl) {
        CVK_ABL(0) CVK_ABL(1) CVK_ABL(2) CVK_ABL(3) CVK_ABL(4) CVK_ABL(7) CVK_ABL(8) CVK_ABL(9) CVK_ABL(11) CVK_ABL(15) CVK_ABL(16) CVK_ABL(24) CVK_ABL(31)
        default: return -1;
    }
#undef CVK_ABL
    CVK_LAUNCH_RETURN("cvk_conv3x3_wino4f_ablate");
}
// the same switches on the fp16 split-operand form (tools/bench_wino4h_ablate.sh)
extern "C" int cvk_conv3x3_wino4h_ablate(const float* x, const void* Uh, const float* bias, float* y, const void* amax_x, const void* amax_w, int N,
                                         int H, int W, int Cin, int Cout, int ldy, int abl, void* stream) {
    const int Wt = (W + 3) / 4, Mt = N * H * Wt, Mpix = N * H * W;
    const int tilesN = f_tiles_n(Cout), tilesM = cvk_cdiv(Mt, F_BM), ntiles = tilesM * tilesN;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipStream_t s = (hipStream_t)stream;
    const FSplit sp = {(const unsigned*)amax_x, (const unsigned*)amax_w, cvk_split_tab(4, CVK_SPLIT_KIND_B), cvk_split_tab(4, CVK_SPLIT_KIND_G)};
#define CVK_ABLH(A) case A: hipLaunchKernelGGL((k_conv3x3_wino4f<false, A, false, true>), dim3(grid), dim3(512), 0, s, x, (const float*)Uh, bias, y, nullptr, nullptr, Mt, H, W, Wt, Cin, Cout, ldy, tilesN, ntiles, Mpix, tilesM, FBnRed{nullptr, nullptr, nullptr, nullptr, nullptr}, sp, FVpl{nullptr, 0, 0}); break;
    switch (abl) {
        CVK_ABLH(0) CVK_ABLH(1) CVK_ABLH(2) CVK_ABLH(4) CVK_ABLH(8) CVK_ABLH(9) CVK_ABLH(11) CVK_ABLH(15) CVK_ABLH(16) CVK_ABLH(31) CVK_ABLH(32) CVK_ABLH(33) CVK_ABLH(64) CVK_ABLH(65)
        default: return -1;
    }
#undef CVK_ABLH
    CVK_LAUNCH_RETURN("cvk_conv3x3_wino4h_ablate");
}
#endif
