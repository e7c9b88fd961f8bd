// split3.hip — the GEMMs of the OPT-IN split-operand modes (cvk.set_split_operands / runner.w2d_split; the executor's DEFAULT fp32 path is
// exact-fp32 MFMA and does not come here; began as the study VERDICT r4 #8 asked for): the batched GEMM stage of the 2-D Winograd path,
// M_xi[T][Cout] = V_xi[T][Cin] * U_xi[Cout][Cin]^T (csrc/wino2d.hip k_w2d_gemm; reference op: the multiply-adds of nn.Conv2d(3x3) in
// models/unet.py:11) and its weight-grad twin P_xi = E_xi^T V_xi, on the 16-bit matrix pipe with SPLIT fp32 operands.  Two formats
// (csrc/split_fmt.h): three bf16 terms (described below; FMT 3) and two scaled fp16 terms with three cross-products (FMT 2).
//
// Every fp32 value x is written as three bf16 terms x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 8 + 8 + 8
// mantissa bits) and a product x*w as its six largest cross-products x1w1 + x1w2 + x2w1 + x1w3 + x3w1 + x2w2 (the dropped ones are
// below 2^-24 relative).  Each cross-product of two bf16 values is EXACT in fp32 and is accumulated in the fp32 accumulators of
// v_mfma_f32_16x16x32_bf16 — 6 bf16 MFMAs replace 16 cycles-equivalents of fp32 MFMA: 0.375 of the matrix time.  Numerically at least as
// good as the fp32 MFMA path (tools/study/split_bf16_model.py: 8e-7 / 1.8e-6 relative L2 for F(4x4) / F(6x6) at 256 channels against
// 1.5e-6 / 3.0e-6 with fp32 products; on the device: tests/test_gpu_split3.py).
//
// Operand format ("split planes"): bf16 [xi][Cin/32][term 3][Rpad][32], rows padded with zeros to a multiple of 256 (V) / 128 (U), the
// four 16-byte chunks of a 64-byte row stored at position c ^ (2 * ((row >> 2) & 1)) — byte for byte the LDS image of a (term, row tile,
// channel slice) block, so a tile's K loop streams linear 8 KiB ranges by LDS-DMA (the tile-major idea of conv_bf16p.hip).
// cvk_split_planes converts fp32 planes [xi][R][C] into it (a stand-alone pass for tests and studies: the transform kernels of
// csrc/wino2d.hip emit the terms from their own store loops).
//
// GEMM kernel: the ping-pong machine of conv_bf16p.hip.  One workgroup = 8 waves = two groups of four (waves w, w + 4 share a SIMD),
// tile = 256 rows of V (128 per group) x 128 rows of U; a wave owns 64 (co) x 64 (t): 16 accumulator blocks, and per 32-channel slice
// 24 fragments (4 blocks x 3 terms per operand, one ds_read_b128 each) feed 96 MFMAs — one slice is one phase: group A's MFMA phase
// runs beside group B's LOAD phase (fragment reads of ITS slice, DMA issue for the slice after), then the roles swap.  Two LDS stages
// of 72 KiB (V terms 48 + U terms 24); a group requests slice k + 1 at the start of its LOAD(k) and confirms it (vmcnt(0)) at the end
// of its MFMA(k), ~1.5 phases later; group A moves the shared U tile as well.  D = U * V^T orientation (A operand = U): a lane ends
// with four consecutive output channels of one tile row -> 16-byte stores.
#include <type_traits>
#include "cvk_common.h"
#include "lds_dma.h"
#include "split_fmt.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// operand fragments of the two formats (split_fmt.h): FMT 3 = three bf16 terms, six cross-products; FMT 2 = two fp16 terms, three
template <int FMT> struct SplitOps;
template <> struct SplitOps<3> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4v mfma(frag a, frag b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct SplitOps<2> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4v mfma(frag a, frag b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
// the cross-products of one K slice, the smallest terms first: f(a term, b term)
template <int FMT, typename F> __device__ __forceinline__ void split_products(F f) {
    if (FMT == 3) { f(2, 0); f(0, 2); f(1, 1); f(1, 0); f(0, 1); f(0, 0); }
    else { f(1, 0); f(0, 1); f(0, 0); }
}

__device__ __forceinline__ unsigned bf16_bits(float v) {          // round to nearest even
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_val(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// fp32 planes [NX][R][C] -> split planes [NX][C/32][3][Rpad][32] (see the file header); one thread = one 8-channel chunk of one row
template <int FMT>
__global__ __launch_bounds__(256) void k_split3_planes(const float* __restrict__ P, unsigned* __restrict__ S, int R, int Rpad, int C, long total,
                                                      const unsigned* __restrict__ amax, CvkSplitTab tab) {
    const int c8n = C >> 3, ncs = C >> 5;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        const long xr = i / c8n;
        const int r = (int)(xr % Rpad), xi = (int)(xr / Rpad);
        unsigned t[3][4] = {};
        if (r < R && FMT == 2) {
            const float sc = cvk_pow2f(cvk_split_exp_xi(amax, tab, xi));
            const f32x4 lo = *reinterpret_cast<const f32x4*>(P + ((size_t)xi * R + r) * C + c8 * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(P + ((size_t)xi * R + r) * C + c8 * 8 + 4);
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[e] * sc;
                const _Float16 h1 = (_Float16)x;
                const _Float16 h2 = (_Float16)(x - (float)h1);
                const int sh = (e & 1) * 16;
                t[0][e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, h1) << sh;
                t[1][e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, h2) << sh;
            }
        } else if (r < R) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(P + ((size_t)xi * R + r) * C + c8 * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(P + ((size_t)xi * R + r) * C + c8 * 8 + 4);
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned b1 = bf16_bits(v[e]);
                const float r1 = v[e] - bf16_val(b1);
                const unsigned b2 = bf16_bits(r1);
                const unsigned b3 = bf16_bits(r1 - bf16_val(b2));
                const int sh = (e & 1) * 16;
                t[0][e >> 1] |= b1 << sh; t[1][e >> 1] |= b2 << sh; t[2][e >> 1] |= b3 << sh;
            }
        }
        const int cs = c8 >> 2, pos = (c8 & 3) ^ (((r >> 2) & 1) << 1);
#pragma unroll
        for (int k = 0; k < FMT; ++k) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 o = {t[k][0], t[k][1], t[k][2], t[k][3]};
            *reinterpret_cast<u32x4*>(S + ((((size_t)xi * ncs + cs) * FMT + k) * Rpad + r) * 16 + pos * 4) = o;
        }
    }
}

template <int IMM> __device__ __forceinline__ void dma16_s(unsigned voff, const void* sbase, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM) : "memory", "scc");
}
__device__ __forceinline__ void pbar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
template <typename FR> __device__ __forceinline__ FR rd16(const char* p) { return *reinterpret_cast<const FR*>(p); }

constexpr int S3_TM = 256, S3_TN = 128;                 // tile: rows of V (tile index t) x rows of U (output channels)

// DBG (experiments build only, tools/tile_stamps_split.py): s_memtime stamps of waves 0 and 4 of workgroup `dbg_wg` -> dbg[2][64]
template <int FMT, bool DBG = false>
__global__ __launch_bounds__(512, 2) void k_gemm_split3(const char* __restrict__ V3, const char* __restrict__ U3, float* __restrict__ Mo,
                                                       int T, int Tpad, int Cin, int Cout, int Cpad, int tilesM, int tilesN,
                                                       const unsigned* __restrict__ amaxV, const unsigned* __restrict__ amaxU, CvkSplitTab tabV,
                                                       CvkSplitTab tabU, unsigned long long* __restrict__ dbg = nullptr, int dbg_wg = 0) {
    int nstamp = 0;
    auto stamp = [&]() {
        if (DBG && (int)blockIdx.x == dbg_wg && (threadIdx.x & 255) == 0 && nstamp < 64) dbg[(threadIdx.x >> 8) * 64 + nstamp++] = __builtin_amdgcn_s_memtime();
    };
    stamp();
    typedef SplitOps<FMT> OPS;
    typedef typename OPS::frag FR;
    constexpr int S3_XT = FMT * S3_TM * 64;                 // 48 (32) KiB: the V terms of a slice
    constexpr int S3_WT = FMT * S3_TN * 64;                 // 24 (16) KiB: the U terms
    constexpr int S3_STAGE = S3_XT + S3_WT;                 // 72 (48) KiB
    // stages: two of 72 KiB for three terms; THREE of 48 KiB for two terms — a two-term MFMA phase lasts ~0.4 us, so a slice requested 1.5
    // phases ahead (two stages) has not crossed the fabric when it is wanted; with three stages it is requested 3.5 phases ahead and the wait
    // at the end of an MFMA phase leaves the youngest slice in flight (counted vmcnt)
    constexpr int NST = FMT == 2 ? 3 : 2;
    __shared__ __attribute__((aligned(1024))) char smem[NST * S3_STAGE];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int grp = wave >> 2, wi = wave & 3, wc = wave & 1, wp = (wave >> 1) & 1;
    const int ncs = Cin >> 5;
    // tile order: the Cout tiles of one row tile are neighbours (they re-read the same V rows: same XCD, same L2), then row tiles, then xi
    const int bid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN;
    const int mt = (bid / tilesN) % tilesM;
    const int xi = bid / (tilesN * tilesM);
    // FMT 2: both operands carry a power-of-two scale per transform index (split_fmt.h); undoing it is exact.  The exponents come from the two
    // amax blocks: read here, under the first slice's flight, not in the epilogue
    const int unscale_exp = FMT == 2 ? -(cvk_split_exp_xi(amaxV, tabV, xi) + cvk_split_exp_xi(amaxU, tabU, xi)) : 0;

    // ---- DMA: this wave moves pieces wi and wi + 4 (1 KiB each) of every term of its group's 128 V rows; group A also of the U tile
    const size_t xterm = (size_t)Tpad * 64, wterm = (size_t)Cpad * 64;          // bytes between two terms of one slice
    const char* xsrc = V3 + ((size_t)xi * ncs * FMT * Tpad + (size_t)mt * S3_TM + grp * 128) * 64;     // slice 0, term 0, this group's rows
    const char* wsrc = U3 + ((size_t)xi * ncs * FMT * Cpad + (size_t)nt * S3_TN) * 64;
    const unsigned voff = wi * 1024 + lane * 16;
    const unsigned xdst = smem_addr + grp * 8192 + wi * 1024;                   // + stage * S3_STAGE + term * 16384 (+ 4096)
    const unsigned wdst = smem_addr + S3_XT + wi * 1024;                        // + stage * S3_STAGE + term * 8192 (+ 4096)
    auto issue_slice = [&](unsigned stage_off) {                               // requests the slice xsrc / wsrc point at, then advances them
        dma16_s<0>(voff, xsrc, xdst + stage_off);
        dma16_s<4096>(voff, xsrc + 4096, xdst + stage_off);
        dma16_s<16384>(voff, xsrc + xterm, xdst + stage_off);
        dma16_s<16384 + 4096>(voff, xsrc + xterm + 4096, xdst + stage_off);
        if (FMT == 3) {
            dma16_s<32768>(voff, xsrc + 2 * xterm, xdst + stage_off);
            dma16_s<32768 + 4096>(voff, xsrc + 2 * xterm + 4096, xdst + stage_off);
        }
        xsrc += FMT * xterm;
        if (grp == 0) {
            dma16_s<0>(voff, wsrc, wdst + stage_off);
            dma16_s<4096>(voff, wsrc + 4096, wdst + stage_off);
            dma16_s<8192>(voff, wsrc + wterm, wdst + stage_off);
            dma16_s<8192 + 4096>(voff, wsrc + wterm + 4096, wdst + stage_off);
            if (FMT == 3) {
                dma16_s<16384>(voff, wsrc + 2 * wterm, wdst + stage_off);
                dma16_s<16384 + 4096>(voff, wsrc + 2 * wterm + 4096, wdst + stage_off);
            }
            wsrc += FMT * wterm;
        }
    };

    // ---- fragment addresses: U rows (A operand) wc*64 + rb*16 + l15, V rows (B operand) grp*128 + wp*64 + cb*16 + l15; chunk q4 swizzled
    const int sw = (q4 ^ (((l15 >> 2) & 1) << 1)) << 4;
    int wa = S3_XT + (wc * 64 + l15) * 64 + sw;             // + term * 8192 + rb * 1024
    int xa = (grp * 128 + wp * 64 + l15) * 64 + sw;         // + term * 16384 + cb * 1024
    int so = 0;                                 // byte offset of the stage being read
    unsigned dso = 0;                           // ... of the stage the next request goes to
    // DMA instructions of one slice for this wave (what a counted wait leaves in flight)
    auto wait_keep_one_slice = [&]() {
        if (FMT == 3) { if (grp == 0) cvk_wait_vm<12>(); else cvk_wait_vm<6>(); }
        else          { if (grp == 0) cvk_wait_vm<8>(); else cvk_wait_vm<4>(); }
    };

    f32x4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    issue_slice(0);
    dso = S3_STAGE;
    if (NST == 3 && ncs > 1) {
        issue_slice(dso);
        dso = 2 * S3_STAGE;
        wait_keep_one_slice();
    } else {
        cvk_wait_vm<0>();
    }
    pbar();
    stamp();
    if (grp == 1) pbar();                       // group B runs one interval behind group A

    for (int cs = 0; cs < ncs; ++cs) {
        // ======== LOAD phase: request slice cs + NST - 1 into the stage everybody read out two phases ago, read this slice's fragments
        const bool more = cs + NST - 1 < ncs;
        if (more) {
            issue_slice(dso);
            dso = dso + S3_STAGE == (unsigned)(NST * S3_STAGE) ? 0u : dso + S3_STAGE;
        }
        FR w[FMT][4], x[FMT][4];
#pragma unroll
        for (int k = 0; k < FMT; ++k) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) w[k][rb] = rd16<FR>(smem + (wa + so + k * 8192 + rb * 1024));
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) x[k][cb] = rd16<FR>(smem + (xa + so + k * 16384 + cb * 1024));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        pbar();
        stamp();
        // ======== MFMA phase: 6 (3) cross-products x 16 blocks, the smallest terms first
        __builtin_amdgcn_s_setprio(1);
        split_products<FMT>([&](int kw, int kx) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[rb][cb] = OPS::mfma(w[kw][rb], x[kx][cb], acc[rb][cb]);
        });
        so = so + S3_STAGE == NST * S3_STAGE ? 0 : so + S3_STAGE;
        __builtin_amdgcn_s_setprio(0);
        stamp();
        // this wave's pieces of the next slice have landed; with three stages the slice after it (requested in this LOAD phase) stays in flight
        if (NST == 3 && more) wait_keep_one_slice(); else cvk_wait_vm<0>();
        stamp();
        pbar();
    }
    if (grp == 0) pbar();
    stamp();

    // ---- epilogue: acc[rb][cb][j] = M[t = mt*256 + grp*128 + wp*64 + cb*16 + l15][co = nt*128 + wc*64 + rb*16 + 4*q4 + j]
    float* const mo = Mo + (size_t)xi * T * Cout;
    const CvkUnscale un = cvk_unscale(unscale_exp);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int t = mt * S3_TM + grp * 128 + wp * 64 + cb * 16 + l15;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int co = nt * S3_TN + wc * 64 + rb * 16 + 4 * q4;
            if (t < T && co < Cout) *reinterpret_cast<f32x4v*>(mo + (size_t)t * Cout + co) = FMT == 2 ? acc[rb][cb] * un.a * un.b : acc[rb][cb];
        }
    }
    stamp();
}


// ---- two-term format, 128 x 128 tiles, four waves, TWO workgroups per CU ------------------------------------------------------------------------
// Time stamps of the 256 x 128 ping-pong workgroup (tools/tile_stamps_split.py, 256 -> 256 channels: 8 slices): 7-11 thousand cycles until the
// first slice has landed, 2.5 thousand per slice, 4-8 thousand for the 16 stores per lane — with ONE workgroup per CU (96-144 KiB of LDS) 40 % of a
// tile's time is its own start and end, and the channel depth of these layers is only 8-32 slices.  This is what the exact-fp32 GEMM (k_w2d_gemm,
// wino2d.hip) learned in round 2: 128 x 128 tiles of four waves and 64 KiB so that TWO workgroups share a CU and one's prologue / stores run under
// the other's MFMAs.  Same planes, same fragment layout (a wave owns 64 x 64), two stages of 32 KiB, one barrier per slice, no wave groups.
// Measured (profiles/r05_g_split_gemm_timing.txt conditions): 256 -> 128 @180x240 225 -> 195 us, 256 -> 256 @90x120 112 -> 114, 1024 -> 512 @45x60 139 -> 157:
// the two machines meet at the same 6-8 TB/s of operand streaming, so this one only takes the layers with ONE Cout tile (Cpad = 128).
__global__ __launch_bounds__(256, 2) void k_gemm_split2q(const char* __restrict__ V3, const char* __restrict__ U3, float* __restrict__ Mo,
                                                        int T, int Tpad, int Cin, int Cout, int Cpad, int tilesM, int tilesN,
                                                        const unsigned* __restrict__ amaxV, const unsigned* __restrict__ amaxU, CvkSplitTab tabV,
                                                        CvkSplitTab tabU, int probe = 0) {
    typedef SplitOps<2> OPS;
    typedef OPS::frag FR;
    constexpr int BLK = 128 * 64;                           // 8 KiB: one term of one operand of a slice
    constexpr int STAGE = 4 * BLK;                          // V h1 | V h2 | U h1 | U h2
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int wc = wave & 1, wp = wave >> 1;
    const int ncs = Cin >> 5;
    const int bid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN;
    const int mt = (bid / tilesN) % tilesM;
    const int xi = bid / (tilesN * tilesM);
    const int unscale_exp = -(cvk_split_exp_xi(amaxV, tabV, xi) + cvk_split_exp_xi(amaxU, tabU, xi));

    const size_t xterm = (size_t)Tpad * 64, wterm = (size_t)Cpad * 64;
    // probe (experiments build, timing only, WRONG values): 1 = every workgroup reads the operands of tile (xi 0, 0, 0): all loads hit the L2;
    // 2 = additionally no stores
    const char* xsrc = V3 + (probe ? 0 : ((size_t)xi * ncs * 2 * Tpad + (size_t)mt * 128) * 64);
    const char* wsrc = U3 + (probe ? 0 : ((size_t)xi * ncs * 2 * Cpad + (size_t)nt * 128) * 64);
    const unsigned voff = wave * 1024 + lane * 16;          // this wave moves pieces wave and wave + 4 of each of the four blocks
    const unsigned dst = smem_addr + wave * 1024;
    auto issue_slice = [&](unsigned stage_off) {
        dma16_s<0>(voff, xsrc, dst + stage_off);
        dma16_s<4096>(voff, xsrc + 4096, dst + stage_off);
        dma16_s<BLK>(voff, xsrc + xterm, dst + stage_off);
        dma16_s<BLK + 4096>(voff, xsrc + xterm + 4096, dst + stage_off);
        dma16_s<2 * BLK>(voff, wsrc, dst + stage_off);
        dma16_s<2 * BLK + 4096>(voff, wsrc + 4096, dst + stage_off);
        dma16_s<3 * BLK>(voff, wsrc + wterm, dst + stage_off);
        dma16_s<3 * BLK + 4096>(voff, wsrc + wterm + 4096, dst + stage_off);
        xsrc += 2 * xterm;
        wsrc += 2 * wterm;
    };
    const int sw = (q4 ^ (((l15 >> 2) & 1) << 1)) << 4;
    const int xa = (wp * 64 + l15) * 64 + sw;               // + term * BLK + cb * 1024
    const int wa = 2 * BLK + (wc * 64 + l15) * 64 + sw;     // + term * BLK + rb * 1024

    f32x4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    issue_slice(0);
    cvk_wait_vm<0>();
    pbar();
    int so = 0;
    for (int cs = 0; cs < ncs; ++cs) {
        if (cs + 1 < ncs) issue_slice(so ? 0 : STAGE);      // the other stage: every wave read it out before the barrier that ended slice cs - 1
        FR w[2][4], x[2][4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) w[k][rb] = rd16<FR>(smem + (wa + so + k * BLK + rb * 1024));
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) x[k][cb] = rd16<FR>(smem + (xa + so + k * BLK + cb * 1024));
        }
        split_products<2>([&](int kw, int kx) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[rb][cb] = OPS::mfma(w[kw][rb], x[kx][cb], acc[rb][cb]);
        });
        so = so ? 0 : STAGE;
        cvk_wait_vm<0>();               // this wave's pieces of the next slice have landed
        cvk_lds_retire_barrier();       // ... everybody's, and everybody has read this slice out
    }

    float* const mo = Mo + (size_t)xi * T * Cout;
    const CvkUnscale un = cvk_unscale(unscale_exp);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int t = mt * 128 + wp * 64 + cb * 16 + l15;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int co = nt * 128 + wc * 64 + rb * 16 + 4 * q4;
            if (t < T && co < Cout && probe < 2) *reinterpret_cast<f32x4v*>(mo + (size_t)t * Cout + co) = acc[rb][cb] * un.a * un.b;
        }
    }
}

// ---- the weight-grad GEMM on split planes: P_xi[co][ci] = sum_t E_xi[t][co] * V_xi[t][ci]  (k = the tile index t) ----------------------------
// Both operands are the split planes the transforms wrote ([xi][C/32][term][Tpad][32]: rows = t) — the SAME V planes the forward GEMM read,
// kept for the backward pass, and E = A dy A^T from the dy transform.  The MFMA wants 8 consecutive k (= t) per lane for its channel:
// ds_read_b64_tr_b16 (lane group = 4 rows x 16 channels, a lane receives its channel's 4 rows; two reads = 8 rows), as k_wgrad_bf16r reads
// its pixel-major tiles.  Same ping-pong machine as k_gemm_split3: 8 waves, two groups, tile 256 x 128 with the groups splitting the
// 256 side (CO256: output channels, else input channels), a wave owns 64 (ci) x 64 (co): per 32-row slice 24 fragments = 48 transposing
// reads and 96 MFMAs; two stages of 72 KiB; the A operand is V (m = ci) so that a lane ends with 4 consecutive input channels of one output
// channel: 16-byte stores into P [part][xi][Cout][Cin].  The depth is cut into f ranges (k_w2d_wgrad_out adds the planes in a fixed order).
template <bool CO256, int FMT>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_split3(const char* __restrict__ E3, const char* __restrict__ V3, float* __restrict__ P,
                                                          int Tpad, int Cin, int Cout, int tilesCi, int tilesCo, int f, int NX,
                                                          const unsigned* __restrict__ amaxE, const unsigned* __restrict__ amaxV, CvkSplitTab tabE,
                                                          CvkSplitTab tabV) {
    typedef SplitOps<FMT> OPS;
    typedef typename OPS::frag FR;
    constexpr int X256 = 8 * FMT * 2048, X128 = 4 * FMT * 2048;       // 48 (32) KiB + 24 (16) KiB per stage
    constexpr int STAGE = X256 + X128;
    constexpr int NST = FMT == 2 ? 3 : 2;                             // as k_gemm_split3: three stages of 48 KiB for two terms
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4, tq = l15 >> 2, tp = l15 & 3;
    const int grp = wave >> 2, wi = wave & 3, wa_ = wave & 1, wb_ = (wave >> 1) & 1;
    int bid = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int part = bid % f; bid /= f;
    const int tci = bid % tilesCi; bid /= tilesCi;
    const int tco = bid % tilesCo;
    const int xi = bid / tilesCo;
    const int unscale_exp = FMT == 2 ? -(cvk_split_exp_xi(amaxE, tabE, xi) + cvk_split_exp_xi(amaxV, tabV, xi)) : 0;
    const int nK = Tpad >> 5;
    const int k0 = (int)((long)nK * part / f), k1 = (int)((long)nK * (part + 1) / f);
    const int ncsE = Cout >> 5, ncsV = Cin >> 5;
    // the 256 side ("big") and the 128 side ("small"): slices of 32 channels
    const char* const big = CO256 ? E3 : V3;
    const char* const sml = CO256 ? V3 : E3;
    const int ncsB = CO256 ? ncsE : ncsV, ncsS = CO256 ? ncsV : ncsE;
    const int sB0 = (CO256 ? tco : tci) * 8, sS0 = (CO256 ? tci : tco) * 4;
    const size_t term = (size_t)Tpad * 64;                           // bytes between two terms of a slice
    // this wave moves slice (4 grp + wi) of the big side and (group A) slice wi of the small side: 3 terms x 2 KiB each, per K slice
    const char* bsrc = big + (((size_t)xi * ncsB + sB0 + 4 * grp + wi) * FMT * Tpad + (size_t)k0 * 32) * 64;
    const char* ssrc = sml + (((size_t)xi * ncsS + sS0 + wi) * FMT * Tpad + (size_t)k0 * 32) * 64;
    const unsigned voff = lane * 16;
    const unsigned bdst = smem_addr + (4 * grp + wi) * FMT * 2048;
    const unsigned sdst = smem_addr + X256 + wi * FMT * 2048;
    auto issue_slice = [&](unsigned stage_off) {
        dma16_s<0>(voff, bsrc, bdst + stage_off);
        dma16_s<1024>(voff, bsrc + 1024, bdst + stage_off);
        dma16_s<2048>(voff, bsrc + term, bdst + stage_off);
        dma16_s<2048 + 1024>(voff, bsrc + term + 1024, bdst + stage_off);
        if (FMT == 3) {
            dma16_s<4096>(voff, bsrc + 2 * term, bdst + stage_off);
            dma16_s<4096 + 1024>(voff, bsrc + 2 * term + 1024, bdst + stage_off);
        }
        bsrc += 2048;
        if (grp == 0) {
            dma16_s<0>(voff, ssrc, sdst + stage_off);
            dma16_s<1024>(voff, ssrc + 1024, sdst + stage_off);
            dma16_s<2048>(voff, ssrc + term, sdst + stage_off);
            dma16_s<2048 + 1024>(voff, ssrc + term + 1024, sdst + stage_off);
            if (FMT == 3) {
                dma16_s<4096>(voff, ssrc + 2 * term, sdst + stage_off);
                dma16_s<4096 + 1024>(voff, ssrc + 2 * term + 1024, sdst + stage_off);
            }
            ssrc += 2048;
        }
    };
    // transposing fragment reads: block of 16 channels = half h16 of slice sl; the lane addresses rows 8 q4 + tq (first read) and + 4 (second:
    // its chunk sits at position ^ 2, the swizzle bit of rows 4..7), channels 4 tp .. 4 tp + 3 of the block
    typedef short s16x4t __attribute__((ext_vector_type(4)));
    // per-lane byte offsets inside a (slice, term) block of 2 KiB for the two reads of the two 16-channel halves: [h16][read]
    int fo[2][2];
#pragma unroll
    for (int h16 = 0; h16 < 2; ++h16) {
        const int chunk = h16 * 2 + (tp >> 1);
        fo[h16][0] = (8 * q4 + tq) * 64 + (chunk << 4) + (tp & 1) * 8;
        fo[h16][1] = (8 * q4 + tq + 4) * 64 + ((chunk ^ 2) << 4) + (tp & 1) * 8;
    }
    auto rd = [&](int blk_off, int h16) {
        const s16x4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(smem + blk_off + fo[h16][0]));
        const s16x4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(smem + blk_off + fo[h16][1]));
        return __builtin_bit_cast(FR, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // wave tile: 64 ci x 64 co.  On the 256 side the group takes 128 channels (4 slices) and the wave 64 of them (2 slices); on the 128 side
    // the wave takes 64 (2 slices).  ci is the A operand.
    const int ciS = CO256 ? 2 * wa_ : 4 * grp + 2 * wa_;       // first ci slice of this wave inside its LDS region
    const int coS = CO256 ? 4 * grp + 2 * wb_ : 2 * wb_;
    const int ciReg = CO256 ? X256 : 0, coReg = CO256 ? 0 : X256;      // which region holds ci / co

    f32x4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    int so = 0;
    unsigned dso = STAGE;
    auto wait_keep_one_slice = [&]() {
        if (FMT == 3) { if (grp == 0) cvk_wait_vm<12>(); else cvk_wait_vm<6>(); }
        else          { if (grp == 0) cvk_wait_vm<8>(); else cvk_wait_vm<4>(); }
    };
    issue_slice(0);
    if (NST == 3 && k0 + 1 < k1) {
        issue_slice(dso);
        dso = 2 * STAGE;
        wait_keep_one_slice();
    } else {
        cvk_wait_vm<0>();
    }
    pbar();
    if (grp == 1) pbar();
    for (int ks = k0; ks < k1; ++ks) {
        const bool more = ks + NST - 1 < k1;
        if (more) {
            issue_slice(dso);
            dso = dso + STAGE == (unsigned)(NST * STAGE) ? 0u : dso + STAGE;
        }
        FR A[FMT][4], B[FMT][4];
#pragma unroll
        for (int k = 0; k < FMT; ++k)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                A[k][mb] = rd(so + ciReg + ((ciS + (mb >> 1)) * FMT + k) * 2048, mb & 1);
                B[k][mb] = rd(so + coReg + ((coS + (mb >> 1)) * FMT + k) * 2048, mb & 1);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pbar();
        __builtin_amdgcn_s_setprio(1);
        split_products<FMT>([&](int ka, int kb) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb][nb] = OPS::mfma(A[ka][mb], B[kb][nb], acc[mb][nb]);
        });
        __builtin_amdgcn_s_setprio(0);
        so = so + STAGE == NST * STAGE ? 0 : so + STAGE;
        if (NST == 3 && more) wait_keep_one_slice(); else cvk_wait_vm<0>();
        pbar();
    }
    if (grp == 0) pbar();

    // acc[mb][nb][j]: ci = ci0 + mb*16 + 4 q4 + j, co = co0 + nb*16 + l15
    const int ci0 = tci * (CO256 ? 128 : 256) + (CO256 ? 64 * wa_ : 128 * grp + 64 * wa_);
    const int co0 = tco * (CO256 ? 256 : 128) + (CO256 ? 128 * grp + 64 * wb_ : 64 * wb_);
    float* const pp = P + ((size_t)part * NX + xi) * Cout * Cin;
    const CvkUnscale un = cvk_unscale(unscale_exp);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int co = co0 + nb * 16 + l15;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ci = ci0 + mb * 16 + 4 * q4;
            if (co < Cout && ci < Cin) *reinterpret_cast<f32x4v*>(pp + (size_t)co * Cin + ci) = FMT == 2 ? acc[mb][nb] * un.a * un.b : acc[mb][nb];
        }
    }
}

}  // namespace

extern "C" int cvk_split3_rows_pad(int R, int mult) { return (R > 0 && mult > 0) ? cvk_cdiv(R, mult) * mult : 0; }

#define CVK_SPLIT_FMT_OK(who) \
    CVK_CHECK_ARG((tile == 4 || tile == 6) && (fmt == 3 || fmt == 2), "%s: tile is 4 or 6, fmt 3 (bf16 x 3) or 2 (fp16 x 2)", who)

// P fp32 [NX][R][C] -> S 16-bit [NX][C/32][fmt][Rpad][32] (Rpad = cvk_split3_rows_pad(R, 256) for V, (R, 128) for U; rows >= R zero); fmt 2: scaled
// per transform index as a plane of `kind` (split_fmt.h) of a tensor whose cvk_absmax_f32 word is amax.  A stand-alone pass (tests, studies:
// the executor's transforms write split planes themselves)
extern "C" int cvk_split_planes(int fmt, int tile, int kind, const float* P, void* S, const void* amax, int NX, int R, int Rpad, int C, void* stream) {
    const char* who = "cvk_split_planes";
    CVK_SPLIT_FMT_OK(who);
    CVK_CHECK_ARG(P && S && NX > 0 && R > 0 && Rpad >= R && C > 0 && C % 32 == 0 && kind >= 0 && kind <= 2, "%s: bad arguments (C must be a multiple of 32)", who);
    CVK_CHECK_ARG(fmt == 3 || (amax != nullptr && NX == (tile + 2) * (tile + 2)), "%s: fmt 2 needs amax and NX = (tile + 2)^2", who);
    CVK_CHECK_ARG(cvk_aligned16(P) && cvk_aligned16(S), "%s: pointers must be 16-byte aligned", who);
    const long total = (long)NX * Rpad * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    const CvkSplitTab tab = cvk_split_tab(tile, kind);
    if (fmt == 3) hipLaunchKernelGGL(k_split3_planes<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, (unsigned*)S, R, Rpad, C, total, (const unsigned*)amax, tab);
    else hipLaunchKernelGGL(k_split3_planes<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, (unsigned*)S, R, Rpad, C, total, (const unsigned*)amax, tab);
    CVK_LAUNCH_RETURN(who);
}
extern "C" int cvk_split3_planes(const float* P, void* S, int NX, int R, int Rpad, int C, void* stream) {
    return cvk_split_planes(3, 4, 0, P, S, nullptr, NX, R, Rpad, C, stream);
}

// Mo fp32 [NX][T][Cout] = V * U^T per xi, operands as split planes of format fmt (V rows padded to Tpad % 256 == 0, U rows to Cpad % 128 == 0);
// fmt 2: amax_v / amax_u = the cvk_absmax_f32 words the two transforms scaled with (V: B table, U: G table)
extern "C" int cvk_w2d_gemm_split(int fmt, int tile, const void* V, const void* U, float* Mo, const void* amax_v, const void* amax_u, int NX, int T,
                                  int Tpad, int Cin, int Cout, int Cpad, void* stream) {
    const char* who = "cvk_w2d_gemm_split";
    CVK_SPLIT_FMT_OK(who);
    CVK_CHECK_ARG(V && U && Mo && NX > 0 && T > 0 && Cin > 0 && Cout > 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(fmt == 3 || (amax_v && amax_u && NX == (tile + 2) * (tile + 2)), "%s: fmt 2 needs both amax blocks and NX = (tile + 2)^2", who);
    CVK_CHECK_ARG(Cin % 32 == 0 && Cout % 4 == 0 && Tpad % S3_TM == 0 && Tpad >= T && Cpad % S3_TN == 0 && Cpad >= Cout,
                  "%s: Cin %% 32, Cout %% 4, Tpad %% 256, Cpad %% 128", who);
    CVK_CHECK_ARG(cvk_aligned16(V) && cvk_aligned16(U) && cvk_aligned16(Mo), "%s: pointers must be 16-byte aligned", who);
    const int tilesM = Tpad / S3_TM, tilesN = Cpad / S3_TN;
    CVK_CHECK_ARG((long)NX * tilesM * tilesN < (1L << 31), "%s: grid too large", who);
    const CvkSplitTab tV = cvk_split_tab(tile, CVK_SPLIT_KIND_B), tU = cvk_split_tab(tile, CVK_SPLIT_KIND_G);
    const dim3 grid((unsigned)(NX * tilesM * tilesN));
    if (fmt == 3) hipLaunchKernelGGL(k_gemm_split3<3>, grid, dim3(512), 0, (hipStream_t)stream, (const char*)V, (const char*)U, Mo, T, Tpad, Cin, Cout, Cpad,
                                     tilesM, tilesN, (const unsigned*)amax_v, (const unsigned*)amax_u, tV, tU);
    else if (cvk_knob("CVK_SPLIT_Q", Cpad == 128 ? 1 : 0) != 0) {      // 128 x 128 tiles, two workgroups per CU: where it measured faster (one Cout tile)
        const int tm = Tpad / 128;
        hipLaunchKernelGGL(k_gemm_split2q, dim3((unsigned)(NX * tm * tilesN)), dim3(256), 0, (hipStream_t)stream, (const char*)V, (const char*)U, Mo, T, Tpad,
                           Cin, Cout, Cpad, tm, tilesN, (const unsigned*)amax_v, (const unsigned*)amax_u, tV, tU, cvk_knob("CVK_SPLIT_PROBE", 0));
    } else hipLaunchKernelGGL(k_gemm_split3<2>, grid, dim3(512), 0, (hipStream_t)stream, (const char*)V, (const char*)U, Mo, T, Tpad, Cin, Cout, Cpad,
                              tilesM, tilesN, (const unsigned*)amax_v, (const unsigned*)amax_u, tV, tU);
    CVK_LAUNCH_RETURN(who);
}
#ifdef CVK_EXPERIMENTS
// time stamps of one workgroup (tools/tile_stamps_split.py): dbg = 128 uint64
extern "C" int cvk_w2d_gemm_split_dbg(int fmt, int tile, const void* V, const void* U, float* Mo, const void* amax_v, const void* amax_u, int NX, int T,
                                      int Tpad, int Cin, int Cout, int Cpad, void* dbg, int dbg_wg, void* stream) {
    const int tilesM = Tpad / S3_TM, tilesN = Cpad / S3_TN;
    const CvkSplitTab tV = cvk_split_tab(tile, CVK_SPLIT_KIND_B), tU = cvk_split_tab(tile, CVK_SPLIT_KIND_G);
    const dim3 grid((unsigned)(NX * tilesM * tilesN));
    if (fmt == 3) hipLaunchKernelGGL((k_gemm_split3<3, true>), grid, dim3(512), 0, (hipStream_t)stream, (const char*)V, (const char*)U, Mo, T, Tpad, Cin, Cout, Cpad,
                                     tilesM, tilesN, (const unsigned*)amax_v, (const unsigned*)amax_u, tV, tU, (unsigned long long*)dbg, dbg_wg);
    else hipLaunchKernelGGL((k_gemm_split3<2, true>), grid, dim3(512), 0, (hipStream_t)stream, (const char*)V, (const char*)U, Mo, T, Tpad, Cin, Cout, Cpad,
                            tilesM, tilesN, (const unsigned*)amax_v, (const unsigned*)amax_u, tV, tU, (unsigned long long*)dbg, dbg_wg);
    CVK_LAUNCH_RETURN("cvk_w2d_gemm_split_dbg");
}
#endif

extern "C" int cvk_w2d_gemm_split3(const void* V3, const void* U3, float* Mo, int NX, int T, int Tpad, int Cin, int Cout, int Cpad, void* stream) {
    return cvk_w2d_gemm_split(3, 4, V3, U3, Mo, nullptr, nullptr, NX, T, Tpad, Cin, Cout, Cpad, stream);
}

// depth ranges of the split weight-grad GEMM for a layer: enough workgroups to fill the chip twice, at least 4 slices per range
extern "C" int cvk_w2d_gemm_tn_split3_ksplit(int NX, int Tpad, int Cin, int Cout) {
    if (NX <= 0 || Tpad <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const bool co256 = Cout % 256 == 0;
    const int nt = NX * (co256 ? (Cout / 256) * cvk_cdiv(Cin, 128) : cvk_cdiv(Cout, 128) * (Cin / 256));
    const int nK = Tpad / 32;
    int f = cvk_cdiv(512, nt > 0 ? nt : 1);
    if (f > nK / 4) f = nK / 4;
    if (f > 16) f = 16;
    if (f < 1) f = 1;
    return f;
}

// P fp32 [f][NX][Cout][Cin] = E^T V per transform index and depth range, operands as split planes of format fmt with Tpad % 256 == 0 rows; needs
// Cout % 256 == 0 and Cin % 128 == 0, or Cin % 256 == 0 and Cout % 128 == 0; f = cvk_w2d_gemm_tn_split3_ksplit(NX, Tpad, Cin, Cout);
// fmt 2: amax_e / amax_v = the cvk_absmax_f32 words of dy (E: A table) and of x (V: B table)
extern "C" int cvk_w2d_gemm_tn_split(int fmt, int tile, const void* E, const void* V, float* P, const void* amax_e, const void* amax_v, int NX,
                                     int Tpad, int Cin, int Cout, void* stream) {
    const char* who = "cvk_w2d_gemm_tn_split";
    CVK_SPLIT_FMT_OK(who);
    CVK_CHECK_ARG(E && V && P && NX > 0 && Tpad > 0 && Tpad % 256 == 0, "%s: bad arguments", who);
    CVK_CHECK_ARG(fmt == 3 || (amax_e && amax_v && NX == (tile + 2) * (tile + 2)), "%s: fmt 2 needs both amax blocks and NX = (tile + 2)^2", who);
    const bool co256 = Cout % 256 == 0 && Cin % 128 == 0;
    CVK_CHECK_ARG(co256 || (Cin % 256 == 0 && Cout % 128 == 0), "%s: needs Cout %% 256 == 0 and Cin %% 128 == 0, or the reverse", who);
    CVK_CHECK_ARG(cvk_aligned16(E) && cvk_aligned16(V) && cvk_aligned16(P), "%s: pointers must be 16-byte aligned", who);
    const int f = cvk_w2d_gemm_tn_split3_ksplit(NX, Tpad, Cin, Cout);
    const int tilesCo = co256 ? Cout / 256 : Cout / 128, tilesCi = co256 ? Cin / 128 : Cin / 256;
    const dim3 grid((unsigned)(NX * tilesCo * tilesCi * f));
    const CvkSplitTab tE = cvk_split_tab(tile, CVK_SPLIT_KIND_A), tV = cvk_split_tab(tile, CVK_SPLIT_KIND_B);
#define CVK_TN_SPLIT(CO_, F_) hipLaunchKernelGGL((k_gemm_tn_split3<CO_, F_>), grid, dim3(512), 0, (hipStream_t)stream, (const char*)E, (const char*)V, P, Tpad, Cin, Cout, \
                                                 tilesCi, tilesCo, f, NX, (const unsigned*)amax_e, (const unsigned*)amax_v, tE, tV)
    if (co256) { if (fmt == 3) CVK_TN_SPLIT(true, 3); else CVK_TN_SPLIT(true, 2); }
    else       { if (fmt == 3) CVK_TN_SPLIT(false, 3); else CVK_TN_SPLIT(false, 2); }
#undef CVK_TN_SPLIT
    CVK_LAUNCH_RETURN(who);
}
extern "C" int cvk_w2d_gemm_tn_split3(const void* E3, const void* V3, float* P, int NX, int Tpad, int Cin, int Cout, void* stream) {
    return cvk_w2d_gemm_tn_split(3, 4, E3, V3, P, nullptr, nullptr, NX, Tpad, Cin, Cout, stream);
}
