// conv_bf16p.hip — 3x3 convolution on bf16 NHWC tensors, "ping-pong" kernel for the channel-heavy layers of
// BASELINE.json configs[3] (forward of models/unet.py:11 nn.Conv2d(3x3, pad 1) and its data-grad; same arithmetic and
// operand layout in HBM as conv_bf16s.hip: bf16 x bf16 on the matrix cores, fp32 accumulation, BatchNorm statistics partials
// from the fp32 accumulators).
//
// Why a second kernel (round 4; PMC of k_conv_bf16s<128,*>, profiles/r04_a_pmc_mfma_bf16.json): the matrix pipe is busy 51 %
// of the SIMD cycles.  There a wave does everything in turn — DMA issue (~100-180 cycles per 1 KiB piece), twelve fragment
// reads, the counted vmcnt wait, 16 MFMAs, the barrier — and its SIMD partner is a wave of ANOTHER workgroup in a random phase:
// the two collide on the pipe or leave it idle together.  Here the partner is chosen and the alternation is forced:
//   * one workgroup = 8 waves = TWO groups of four (waves w and w+4 share a SIMD), one workgroup per CU;
//   * an interval between two s_barriers is group A's MFMA phase (32 x v_mfma_f32_16x16x32_bf16 = 512 matrix cycles, nothing
//     else in the stream, s_setprio 1) and group B's LOAD phase (its DMA pieces, the twelve ds_read_b128 of ITS next step, the
//     waits), then the roles swap: the pipe of every SIMD always has exactly one wave's MFMAs queued;
//   * both groups multiply the SAME weight tile (128 output channels x 32 input channels of one tap) with pixel rows of their own
//     half of a 16 x 32 pixel tile: a tap's 8 KiB are staged once for 512 pixels (k_conv_bf16s: for 256), 1.6 DMA pieces per wave
//     and step instead of 3;
//   * the weight pack is TILE-MAJOR: the 8 KiB of (output-channel tile, slice, tap) are contiguous and already in LDS image order
//     (swizzle applied by the pack kernel), so a tile's whole K loop streams one linear byte range — 1 KiB-contiguous DMA pieces
//     instead of sixteen 64-byte row fragments, and the address of step s is base + 8192 s;
//   * LDS: two halo slabs (18 x 34 pixels x 64 B, 40 KiB each: slice cs+1 lands during slice cs) + a ring of D+1 weight tiles
//     (D = steps a tile is requested ahead of its first read); counted s_waitcnt vmcnt, raw s_barrier, DMA from inline asm.
// Swizzle (weights: row n; pixels: halo column hx): 16-byte chunk c of a 64-byte row at position c ^ (2 * ((row >> 2) & 1)).
// Environment switches and in-kernel time stamps exist only in the experiments build (cvk_common.h cvk_knob, `make experiments`).
#include <type_traits>
#include "cvk_common.h"
#include "lds_dma.h"
#include "conv_bf16p.h"

namespace {
using namespace cvk_bf16p;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int HP = TW + 2;                      // halo row pitch in pixels
constexpr int SLAB_ROWS = (TH + 2) * HP;        // 612 LDS rows (halo pixels) of 64 B
constexpr int SLAB_PIECES = 40;                 // 16-row DMA pieces per slab: 5 per wave (rows 612..639 come from the zero page)
constexpr int SLAB_BYTES = SLAB_PIECES * 1024;
constexpr int BTAP = BN * 64;                   // 8 KiB: the weight tile of one tap and one channel slice
static_assert(SLAB_PIECES * 16 >= SLAB_ROWS && SLAB_PIECES % 8 == 0, "slab pieces");


__device__ __forceinline__ bf16x8 lds_read16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// sum over the 16 lanes of a DPP row, result in every lane of the row; fixed tree
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    return v;
}

__device__ __forceinline__ void phase_barrier() {
    // phase boundary: nothing moves across it (hipcc otherwise sinks MFMAs below a raw s_barrier and hoists fragment reads above it)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA with the address arithmetic on the SCALAR unit (round-4 finding, tools/stamps_bf16p.py: with per-lane 64-bit addresses
// the v_lshl_add_u64 + global_load_lds pair of a LOAD phase took ~550 cycles while the SIMD partner issued its 16 MFMAs —
// the whole MFMA phase — against ~100 alone; the fragment reads beside it were not delayed).  Weights: 64-bit SGPR base + a
// constant 32-bit lane offset.  Slab: buffer load with LDS destination, per-lane 32-bit byte offset (fixed per piece), the
// channel slice in the scalar offset, out-of-frame lanes get an offset past the end of the image: the range check returns 0.
__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma16_buf(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory");
}

// the same with the LDS destination = scalar base + compile-time offset (one s_add into M0: the dozen destinations of a slice body
// do not each occupy a scalar register)
template <int IMM> __device__ __forceinline__ void dma16_saddr_i(unsigned voff, const void* sbase, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM) : "memory", "scc");
}
template <int IMM> __device__ __forceinline__ void dma16_buf_i(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base), "n"(IMM) : "memory", "scc");
}

// The kernels below run v_mfma_f32_16x16x32_bf16.  MI355X_MICROARCH.md (DVFS give-back, item 7): in MFMA-bound bf16 loops on random data
// the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per FLOP (measured there: 1.12-1.15x the FLOP/s; here
// +7-8 % wall on every layer against round 4's first ping-pong kernel on 32x32x16, which was removed in round 5 — git history: k_conv_bf16p).
// Weights are requested D = 2 steps ahead of their first read into a ring of three tiles: 9 steps per slice, so the slot of a step is a
// compile-time constant of the unrolled slice body and every fragment read is lane base + immediate.  Fragment geometry:
//   * a fragment is 16 rows x 32 k = ONE ds_read_b128 per lane for the whole K slice (lane: row l15 = lane & 15, 16-byte chunk
//     q4 = lane >> 4 of the 64-byte row): 4 weight + 8 pixel fragments and 32 MFMAs (16 cycles each) per step — the same 12 reads
//     and 512 matrix cycles — and ONE lane address per operand and column shift (no k-half variants);
//   * swizzle: chunk c of a row (weights: row n; pixels: halo column hx) at position c ^ (2 * ((row >> 2) & 1)) — conflict-free
//     for the 16-row x 4-chunk read pattern at every column shift (brute-forced over the ds_read_b128 lane groups);
//   * D: lane (l15, q4) holds channels 4 q4 .. 4 q4 + 3 of the 16-channel block for pixel column l15 of the 16-pixel block:
//     32 accumulators of 4 registers; statistics: 16 (sum, sumsq) pairs per lane, the 16 lanes of a DPP row share their channels.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// PERSISTENT (round 4): a workgroup walks the tiles g, g + G, g + 2G, ... (G = grid size = CUs; the XCD remap keeps the tiles that
// run together neighbours).  With one workgroup per CU nothing else covers a tile's prologue (2-4.5 us: the first slab comes from
// HBM) and the launch gap between two workgroups (~1.5 us), against 24 us for the K loop of a 128-channel layer — so the NEXT tile's
// prologue DMAs (slab of slice 0 -> slab buffer A, weights of steps 0 and 1) are issued right after the K loop and land under the
// epilogue.  For that the epilogue keeps out of [0, 64 KiB) (ring + slab A): the output is transposed through a 64 KiB stage at
// [64 KiB, 128 KiB) in TWO passes (group A's 256 pixels x 256 B, then group B's).  Results leave as always-issued buffer stores
// (16 per lane and tile; out-of-frame lanes get an offset past the buffer), so the counted vmcnt of the next tile can step over them.

// t / d for t * d < 2^32 (the launcher checks the tile counts): one multiply-high instead of the ~40 scalar instructions of a division
// by a run-time divisor.  The persistent kernels decode a tile index three times per tile, on every wave, with the matrix pipe idle.
struct TileDiv {
    unsigned m, add_mask, d;
    __device__ explicit TileDiv(int dd) {
        d = (unsigned)dd;
        const unsigned long long mm = 0x100000000ULL / d + 1ULL;
        m = __builtin_amdgcn_readfirstlane((unsigned)mm);
        add_mask = (mm >> 32) ? 0xFFFFFFFFu : 0u;
    }
    __device__ __forceinline__ int div(int t) const { return (int)(__umulhi((unsigned)t, m) + ((unsigned)t & add_mask)); }
    __device__ __forceinline__ int mod(int t, int q) const { return t - q * (int)d; }
};

template <bool STATS>
__global__ __launch_bounds__(512, 2) void k_conv_bf16q(const __bf16* __restrict__ X, const char* __restrict__ Wp,
                                                      const float* __restrict__ bias, __bf16* __restrict__ Y,
                                                      float* __restrict__ stats, float* __restrict__ cnt, int H, int W, int Cin,
                                                      int Cout, int ldy, int tilesX, int tilesY, int tilesN, int P, int ntiles, int nts) {
    constexpr int D = 2, RING = 3;
    constexpr int RING_BYTES = RING * BTAP;                       // [0, 24 KiB) weight ring, [24, 64) slab A, [64, 104) slab B
    constexpr int STAGE_OFF = RING_BYTES + SLAB_BYTES;            // epilogue stage [64 KiB, 128 KiB): 256 pixels x 256 B per pass
    constexpr int STAGE_BYTES = (TH / 2) * TW * BN * 2;
    constexpr int RED_OFF = STAGE_OFF + STAGE_BYTES;              // statistics partials [channel 128][partial 4] (sum, sumsq): 4 KiB
    constexpr int LDS_BYTES = RED_OFF + (STATS ? BN * 4 * 8 : 0);
    constexpr int NSTORE = 16;                                    // buffer stores per lane and tile
    static_assert(STAGE_OFF >= RING_BYTES + SLAB_BYTES && LDS_BYTES >= RING_BYTES + 2 * SLAB_BYTES && LDS_BYTES <= 160 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const unsigned smem_addr = cvk_lds_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int grp = wave >> 2;
    const int wc = wave & 1, wp = (wave >> 1) & 1;
    const int row0 = grp * 8 + wp * 4;
    const int ncs = Cin / CK, nsteps = ncs * 9;
    const int G = gridDim.x;

    // ---- per-tile state: the tile being multiplied (cur_*) and the DMA sources of the tile being requested ------------------------
    struct Geo { int nt, sp, x0, y0, img; };
    const TileDiv divN(tilesN), divX(tilesX), divXY(tilesX * tilesY);
    auto geo_of = [&](int t) {
        Geo g;
        g.sp = divN.div(t);
        g.nt = divN.mod(t, g.sp);
        g.img = divXY.div(g.sp);
        const int rem = divXY.mod(g.sp, g.img);
        const int ty = divX.div(rem), tx = divX.mod(rem, ty);
        g.x0 = tx * TW; g.y0 = ty * TH;
        return g;
    };
    unsigned aoff[5];
    i32x4 xrsrc;
    xrsrc[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)(H * W * Cin) * 2u));
    xrsrc[3] = 0x00020000;
    const unsigned wvoff = wave * 1024 + lane * 16;
    const unsigned wave_lds = smem_addr + wave * 1024;         // LDS destination of this wave's piece of a weight tile / slab group
    const char* wnext = Wp;
    int slab_yx[5];                                           // halo coordinates of the LDS row a lane fills in slab piece 8t + wave
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int row = (8 * t + wave) * 16 + (lane >> 2);
        const int hy = row / HP;
        slab_yx[t] = row < SLAB_ROWS ? (hy << 8) | (row - hy * HP) : 0x4000;       // rows past the slab: far outside every frame
    }
    auto setup_dma = [&](const Geo& g) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int hy = slab_yx[t] >> 8, hx = slab_yx[t] & 255;
            const int chunk = (lane & 3) ^ (((hx >> 2) & 1) << 1);
            const int iy = g.y0 - 1 + hy, ix = g.x0 - 1 + hx;
            const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            aoff[t] = ok ? (unsigned)((iy * W + ix) * Cin + chunk * 8) * 2u : 0x80000000u;
        }
        const uintptr_t xbase = (uintptr_t)(X + (size_t)g.img * H * W * Cin);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xbase);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32) & 0xFFFF);
        wnext = Wp + (size_t)g.nt * nsteps * BTAP;
    };
    // slab piece t of this wave into the slab whose first byte (+ this wave's 1 KiB) is slab_lds; weights of the next step into ring slot SLOT
    auto dma_slab_piece = [&](auto t_tag, int cs, unsigned slab_lds) {
        constexpr int T = decltype(t_tag)::value;
        dma16_buf_i<T * 8192>(aoff[T], xrsrc, (unsigned)cs * (CK * 2), slab_lds);
    };
    auto dma_weights_next = [&](auto slot_tag, int s) {
        dma16_saddr_i<decltype(slot_tag)::value * BTAP>(wvoff, wnext, wave_lds);
        if (s < nsteps - 1) wnext += BTAP;
    };
    auto issue_prologue = [&]() {
        dma_slab_piece(std::integral_constant<int, 0>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 1>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 2>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 3>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 4>{}, 0, wave_lds + RING_BYTES);
        dma_weights_next(std::integral_constant<int, 0>{}, 0);
    };

    // weights: row wc*64 + rb*16 + l15, chunk q4; pixels: halo row row0 + tp + dy, halo column half*16 + l15 + dx, chunk q4
    const int wa = (wc * 64 + l15) * 64 + ((q4 ^ (((l15 >> 2) & 1) << 1)) << 4);
    int pb0[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) pb0[dx] = RING_BYTES + row0 * (HP * 64) + (l15 + dx) * 64 + ((q4 ^ ((((l15 + dx) >> 2) & 1) << 1)) << 4);

    // round k runs the tiles [k G, k G + n_k), n_k = min(G, tiles left), on the first n_k workgroups; the XCD remap is applied per
    // round, so a partial last round is still spread over all eight XCDs (with one remap of the whole grid it ran on half of them:
    // the 384-tile layers lost 6 %)
    int base = 0;
    int tile = cvk_xcd_remap(blockIdx.x, min(G, ntiles));
    Geo cur = geo_of(tile);
    setup_dma(cur);
    issue_prologue();
    bool stores_in_flight = false;

    while (true) {
        // The weights of step 1 are requested HERE, behind the previous tile's stores, not with the prologue in front of them (round 5):
        // the in-order counter then needs no special case in the K loop — vmcnt(2) at the end of step 0 covers the stores as well; the
        // `step == 0 && stores_in_flight` branch inside the first step made hipcc peel the first slice (two copies of the MFMA stream; the
        // data-grad variant spilled 12 registers around them: 3.57 -> 3.49 ms for its 14 launches of a configs[3] step).
        dma_weights_next(std::integral_constant<int, 1>{}, 1);
        // slab of slice 0 and the weights of step 0 have landed; behind them in the queue: (after the first tile) the previous tile's
        // NSTORE stores, and the piece just requested
        if (stores_in_flight) cvk_wait_vm<1 + NSTORE>(); else cvk_wait_vm<1>();
        phase_barrier();
        if (grp == 1) phase_barrier();

        f32x4v acc[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
        int pb[3] = {pb0[0], pb0[1], pb0[2]};
        int pb_flip = SLAB_BYTES;

        int step = 0;
        for (int cs = 0; cs < ncs; ++cs) {
            const unsigned slab_next = wave_lds + RING_BYTES + ((cs + 1) & 1) * SLAB_BYTES;
            const int csn = min(cs + 1, ncs - 1);
            auto step_body = [&](auto sidx_tag) {
                constexpr int sidx = decltype(sidx_tag)::value;
                // ======== LOAD phase
                dma_weights_next(std::integral_constant<int, (sidx + D) % RING>{}, step + D);
                if (sidx < 5) dma_slab_piece(std::integral_constant<int, sidx < 5 ? sidx : 0>{}, csn, slab_next);
                constexpr int dy = sidx / 3, dx = sidx % 3, slot = sidx % RING;
                bf16x8 a[4], b[8];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) a[rb] = lds_read16(smem + (wa + slot * BTAP + rb * 16 * 64));
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) b[cb] = lds_read16(smem + (pb[dx] + ((cb >> 1) + dy) * (HP * 64) + (cb & 1) * 16 * 64));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // this wave's piece of the weights of step + 1 has landed: everything requested before this phase's DMAs — except in
                // the first phase of a later tile, whose step-1 weights sit in front of the previous tile's stores
                if (sidx < 5) cvk_wait_vm<2>();
                else cvk_wait_vm<1>();
                phase_barrier();
                // ======== MFMA phase
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
                if (sidx == 8) {
#pragma unroll
                    for (int dx2 = 0; dx2 < 3; ++dx2) pb[dx2] += pb_flip;
                    pb_flip = -pb_flip;
                }
                __builtin_amdgcn_s_setprio(0);
                phase_barrier();
                ++step;
            };
            step_body(std::integral_constant<int, 0>{}); step_body(std::integral_constant<int, 1>{}); step_body(std::integral_constant<int, 2>{});
            step_body(std::integral_constant<int, 3>{}); step_body(std::integral_constant<int, 4>{}); step_body(std::integral_constant<int, 5>{});
            step_body(std::integral_constant<int, 6>{}); step_body(std::integral_constant<int, 7>{}); step_body(std::integral_constant<int, 8>{});
        }
        if (grp == 0) phase_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the tail re-loads have landed: ring and slabs are free
        phase_barrier();

        // this tile's bias values BEFORE the next tile's DMAs go out: hipcc waits vmcnt(0) for a register load, i.e. for every DMA
        // issued in front of its use as well (tools/tile_stamps_h.py: 2.3 us per tile = the next slab's HBM latency, the very thing the
        // early prologue is there to hide)
        const int n0 = cur.nt * BN;
        f32x4 bvals[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int co = n0 + wc * 64 + rb * 16 + 4 * q4;
            bvals[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && co < Cout) bvals[rb] = *reinterpret_cast<const f32x4*>(bias + co);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) asm volatile("" : "+v"(bvals[rb]));

        // ---- the next tile's prologue goes out before this tile's epilogue ---------------------------------------------------------
        base += G;
        const int n_k = min(G, ntiles - base);
        const bool has_next = (int)blockIdx.x < n_k;
        const int next = has_next ? base + cvk_xcd_remap(blockIdx.x, n_k) : 0;
        Geo nxt = cur;
        if (has_next) {
            nxt = geo_of(next);
            setup_dma(nxt);
            issue_prologue();
        }

        // ---- epilogue: bias, statistics, pack; transposed through the stage in two passes; 16-byte buffer stores -------------------
        // acc[rb][cb][i]: channel n0 + wc*64 + rb*16 + 4*q4 + i, pixel (y0 + row0 + (cb >> 1), x0 + (cb & 1)*16 + l15)
        // the epilogue's lane terms are recomputed per tile from a laundered lane id: hoisted out of the tile loop (they are loop
        // invariant) the ~60 stage / store addresses would live in registers through the K loop and spill
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int l15 = elane & 15, q4 = elane >> 4, lane = elane;

        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Y + ((size_t)(cur.img * H + cur.y0) * W + cur.x0) * ldy + n0), 0, 0x7FFFFFFF, 0x00020000);
        // Every wave: bias, statistics, pack to bf16 IN PLACE (the packed pair of a block replaces the first two of its four
        // accumulator registers: no second register array beside the 128 accumulators).
        // (Round 5: the matrix-pipe statistics of k_conv_bf16h were tried here too — two stage passes, 16 partials per channel — and were 3 %
        // SLOWER on the 12 forward launches, 2.36 -> 2.44 ms: the vector version stays.)
        // Statistics: a lane holds (sum, sum of squares) of 16 channels over its 8 pixel blocks; the 16 lanes of a DPP row hold the
        // same channels for 16 pixel columns: four DPP adds per value leave the row total in every lane; 4 partials per channel
        // (2 groups x 2 row halves) meet in LDS and are combined in fp64 in a fixed order.
        {
            float s[4][4], q[4][4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x4 bv = bvals[rb];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[rb][j] = 0.f; q[rb][j] = 0.f; }
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const bool ok = (cur.x0 + (cb & 1) * 16 + l15 < W) & (cur.y0 + row0 + (cb >> 1) < H);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[rb][cb][j] + bv[j];
                    if (STATS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float vm = ok ? v[j] : 0.f;
                            s[rb][j] += vm;
                            q[rb][j] += vm * vm;
                        }
                    }
                    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    const float2 of = __builtin_bit_cast(float2, o);
                    acc[rb][cb][0] = of.x;
                    acc[rb][cb][1] = of.y;
                    __builtin_amdgcn_sched_barrier(0);      // one block at a time
                }
            }
            if (STATS) {
                float2* const red = reinterpret_cast<float2*>(smem + RED_OFF);
                const int part = grp * 2 + wp;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ss = row_sum16(s[rb][j]), qq = row_sum16(q[rb][j]);
                        if (l15 == 0) red[(wc * 64 + rb * 16 + 4 * q4 + j) * 4 + part] = float2{ss, qq};
                    }
            }
        }
        auto stage_mine = [&]() {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const int p = (wp * 4 + (cb >> 1)) * 32 + (cb & 1) * 16 + l15, chunk = wc * 8 + rb * 2 + (q4 >> 1);
                    *reinterpret_cast<float2*>(smem + STAGE_OFF + p * 256 + ((chunk ^ (p & 15)) << 4) + 8 * (q4 & 1)) = float2{acc[rb][cb][0], acc[rb][cb][1]};
                }
        };
        auto store_pass = [&](int pass) {
            // wave w stores stage rows 32w .. 32w + 31 (= tile row 8*pass + w): 8 instructions of 4 pixels x 256 B
            const int chunk = lane & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = wave * 32 + it * 4 + (lane >> 4);
                const int prow = pass * 8 + (p >> 5), pcol = p & 31;
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + STAGE_OFF + p * 256 + ((chunk ^ (p & 15)) << 4));
                const bool ok = (cur.y0 + prow < H) & (cur.x0 + pcol < W) & (n0 + chunk * 8 < ldy);
                const unsigned off = (unsigned)((prow * W + pcol) * ldy + chunk * 8) * 2u;
                if (nts) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 0);
            }
        };
        if (grp == 0) stage_mine();
        __syncthreads();
        store_pass(0);
        __syncthreads();
        if (grp == 1) stage_mine();
        __syncthreads();
        store_pass(1);
        __syncthreads();
        if (STATS) {
            // every partial was written before the last barrier above
            if (tid < BN) {
                const float2* const red = reinterpret_cast<const float2*>(smem + RED_OFF);
                double S = 0.0, Q = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float2 v = red[tid * 4 + i]; S += (double)v.x; Q += (double)v.y; }
                const int co = n0 + tid;
                const int nvalid = min(TH, H - cur.y0) * min(TW, W - cur.x0);
                if (co < Cout) {
                    const double m2 = Q - S * S / (double)nvalid;
                    stats[(size_t)cur.sp * Cout + co] = (float)S;
                    stats[(size_t)(P + cur.sp) * Cout + co] = (float)(m2 > 0.0 ? m2 : 0.0);
                }
                if (cur.nt == 0 && tid == 0) cnt[cur.sp] = (float)nvalid;
            }
            // the partials are rewritten in the next tile's epilogue, at least four barriers from here
        }
        if (!has_next) break;
        tile = next;
        cur = nxt;
        stores_in_flight = true;
    }
}

#ifdef CVK_EXPERIMENTS
// ---------------------------------------------------------------------------------------------- kernel-column phases (round 5)
// k_conv_bf16q synchronises its two wave groups twice per TAP: 32 MFMAs = 512 matrix cycles between two s_barriers, and the PMC showed
// the waves waiting 35-37 % of their cycles with the matrix pipe busy 64 % (profiles/r04_f_pmc_mfma_bf16.json; VERDICT r4 #1b).  Here a
// phase is a kernel COLUMN — the taps dy = 0, 1, 2 of one dx: 96 MFMAs = 1536 matrix cycles per barrier pair, a third of the barriers.
// The three taps of a column read the same halo rows shifted by one (tile row tp of tap dy = halo row tp + dy), so the 12 pixel
// fragments of halo rows row0 .. row0 + 5 serve all three: 24 fragment reads (96 registers beside the 128 accumulators) per 96 MFMAs
// where three single-tap steps read 36.  The weight stream is packed column-major for it ([slice][dx][dy][128 rows]: k_pack_batch
// `col`), a ring slot = the 24 KiB of a column, slot = dx (compile-time), requested two phases ahead.  Epilogue, tile walk, DMA
// addressing, statistics: k_conv_bf16q's, byte for byte (same accumulator layout).
template <bool STATS>
__global__ __launch_bounds__(512, 2) void k_conv_bf16c(const __bf16* __restrict__ X, const char* __restrict__ Wp,
                                                      const float* __restrict__ bias, __bf16* __restrict__ Y,
                                                      float* __restrict__ stats, float* __restrict__ cnt, int H, int W, int Cin,
                                                      int Cout, int ldy, int tilesX, int tilesY, int tilesN, int P, int ntiles, int nts) {
    // LDS map: ring slot 0 [0, 24) | slot 1 [24, 48) | slab A [48, 88) | slot 2 [88, 112) | slab B [112, 152) KiB.  A ring slot holds the
    // three weight tiles (dy = 0, 1, 2) of one kernel COLUMN of one channel slice, slot = dx.  The next tile's prologue lands in slots 0, 1 and
    // slab A = [0, 88 KiB); slot 2 sits between the slabs so that the epilogue's stage [88, 152) is one contiguous 64 KiB range.
    constexpr int PH = 3 * BTAP;                                  // 24 KiB: the weights of one phase
    constexpr int SLOT0 = 0, SLOT1 = PH, SLAB_A = 2 * PH, SLOT2 = SLAB_A + SLAB_BYTES, SLAB_B = SLOT2 + PH;
    constexpr int STAGE_OFF = SLOT2;                              // epilogue stage [88 KiB, 152 KiB): 256 pixels x 256 B per pass
    constexpr int STAGE_BYTES = (TH / 2) * TW * BN * 2;
    constexpr int RED_OFF = STAGE_OFF + STAGE_BYTES;              // statistics partials [channel 128][partial 4] (sum, sumsq): 4 KiB
    constexpr int LDS_BYTES = RED_OFF + (STATS ? BN * 4 * 8 : 0);
    constexpr int NSTORE = 16;                                    // buffer stores per lane and tile
    static_assert(SLAB_B + SLAB_BYTES == RED_OFF && STAGE_OFF + STAGE_BYTES <= RED_OFF && LDS_BYTES <= 160 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const unsigned smem_addr = cvk_lds_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int grp = wave >> 2;
    const int wc = wave & 1, wp = (wave >> 1) & 1;
    const int row0 = grp * 8 + wp * 4;
    const int ncs = Cin / CK, nph = ncs * 3;
    const int G = gridDim.x;

    // ---- per-tile state: the tile being multiplied (cur_*) and the DMA sources of the tile being requested ------------------------
    struct Geo { int nt, sp, x0, y0, img; };
    const TileDiv divN(tilesN), divX(tilesX), divXY(tilesX * tilesY);
    auto geo_of = [&](int t) {
        Geo g;
        g.sp = divN.div(t);
        g.nt = divN.mod(t, g.sp);
        g.img = divXY.div(g.sp);
        const int rem = divXY.mod(g.sp, g.img);
        const int ty = divX.div(rem), tx = divX.mod(rem, ty);
        g.x0 = tx * TW; g.y0 = ty * TH;
        return g;
    };
    unsigned aoff[5];
    i32x4 xrsrc;
    xrsrc[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)(H * W * Cin) * 2u));
    xrsrc[3] = 0x00020000;
    const unsigned wvoff = wave * 1024 + lane * 16;
    const unsigned wave_lds = smem_addr + wave * 1024;         // LDS destination of this wave's piece of a weight tile / slab group
    const char* wnext = Wp;
    int slab_yx[5];                                           // halo coordinates of the LDS row a lane fills in slab piece 8t + wave
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int row = (8 * t + wave) * 16 + (lane >> 2);
        const int hy = row / HP;
        slab_yx[t] = row < SLAB_ROWS ? (hy << 8) | (row - hy * HP) : 0x4000;       // rows past the slab: far outside every frame
    }
    auto setup_dma = [&](const Geo& g) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int hy = slab_yx[t] >> 8, hx = slab_yx[t] & 255;
            const int chunk = (lane & 3) ^ (((hx >> 2) & 1) << 1);
            const int iy = g.y0 - 1 + hy, ix = g.x0 - 1 + hx;
            const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            aoff[t] = ok ? (unsigned)((iy * W + ix) * Cin + chunk * 8) * 2u : 0x80000000u;
        }
        const uintptr_t xbase = (uintptr_t)(X + (size_t)g.img * H * W * Cin);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xbase);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32) & 0xFFFF);
        wnext = Wp + (size_t)g.nt * nph * PH;
    };
    // slab piece t of this wave into the slab at LDS offset SLAB (+ this wave's 1 KiB); the 24 pieces of the next phase's weights (this wave:
    // pieces wave, wave + 8, wave + 16 = its KiB of the three taps) into ring slot SLOT
    auto dma_slab_piece = [&](auto t_tag, int cs, unsigned slab_lds) {
        constexpr int T = decltype(t_tag)::value;
        dma16_buf_i<T * 8192>(aoff[T], xrsrc, (unsigned)cs * (CK * 2), slab_lds);
    };
    auto dma_phase_next = [&](auto slot_tag, int ph) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int OFF = SLOT == 0 ? SLOT0 : (SLOT == 1 ? SLOT1 : SLOT2);
        dma16_saddr_i<OFF>(wvoff, wnext, wave_lds);
        dma16_saddr_i<OFF + BTAP>(wvoff, wnext + BTAP, wave_lds);
        dma16_saddr_i<OFF + 2 * BTAP>(wvoff, wnext + 2 * BTAP, wave_lds);
        if (ph < nph - 1) wnext += PH;
    };
    auto issue_prologue = [&]() {
        dma_slab_piece(std::integral_constant<int, 0>{}, 0, wave_lds + SLAB_A);
        dma_slab_piece(std::integral_constant<int, 1>{}, 0, wave_lds + SLAB_A);
        dma_slab_piece(std::integral_constant<int, 2>{}, 0, wave_lds + SLAB_A);
        dma_slab_piece(std::integral_constant<int, 3>{}, 0, wave_lds + SLAB_A);
        dma_slab_piece(std::integral_constant<int, 4>{}, 0, wave_lds + SLAB_A);
        dma_phase_next(std::integral_constant<int, 0>{}, 0);
    };

    // weights: row wc*64 + rb*16 + l15, chunk q4; pixels: halo row row0 + tp + dy, halo column half*16 + l15 + dx, chunk q4
    const int wa = (wc * 64 + l15) * 64 + ((q4 ^ (((l15 >> 2) & 1) << 1)) << 4);
    int pb0[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) pb0[dx] = SLAB_A + row0 * (HP * 64) + (l15 + dx) * 64 + ((q4 ^ ((((l15 + dx) >> 2) & 1) << 1)) << 4);

    // round k runs the tiles [k G, k G + n_k), n_k = min(G, tiles left), on the first n_k workgroups; the XCD remap is applied per
    // round, so a partial last round is still spread over all eight XCDs (with one remap of the whole grid it ran on half of them:
    // the 384-tile layers lost 6 %)
    int base = 0;
    int tile = cvk_xcd_remap(blockIdx.x, min(G, ntiles));
    Geo cur = geo_of(tile);
    setup_dma(cur);
    issue_prologue();
    bool stores_in_flight = false;

    while (true) {
        // The weights of phase 1 are requested HERE, behind the previous tile's stores, not with the prologue in front of them: the
        // in-order counter then needs no special case in the K loop (vmcnt(3) at the end of phase 0 covers the stores as well; a
        // stores_in_flight branch inside the first phase made hipcc peel the first slice — two copies of the MFMA stream and spills).
        dma_phase_next(std::integral_constant<int, 1>{}, 1);
        // slab of slice 0 and the weights of phase 0 have landed; behind them in the queue: (after the first tile) the previous tile's
        // NSTORE stores, and the three pieces just requested
        if (stores_in_flight) cvk_wait_vm<3 + NSTORE>(); else cvk_wait_vm<3>();
        phase_barrier();
        if (grp == 1) phase_barrier();

        f32x4v acc[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
        int pb[3] = {pb0[0], pb0[1], pb0[2]};
        int pb_flip = SLAB_B - SLAB_A;

        int ph = 0;
        for (int cs = 0; cs < ncs; ++cs) {
            const unsigned slab_next = wave_lds + (((cs + 1) & 1) ? SLAB_B : SLAB_A);
            const int csn = min(cs + 1, ncs - 1);
            auto col_body = [&](auto dx_tag) {
                constexpr int dx = decltype(dx_tag)::value;
                constexpr int SOFF = dx == 0 ? SLOT0 : (dx == 1 ? SLOT1 : SLOT2);
                // ======== LOAD phase: the weights of phase ph + 2; in the dx = 1 phase the next slice's whole slab (5 pieces); this
                // column's fragments: 12 weight fragments (3 taps x 4 row blocks) and the pixel fragments of halo rows row0 .. row0 + 3.
                // Tile row tp of tap dy reads halo row tp + dy, so the six halo rows row0 .. row0 + 5 serve all three taps (24 reads per 96
                // MFMAs, three single-tap steps read 36); rows 4 and 5 are read DURING the MFMA phase into the registers of rows 0 and 1 once
                // those are dead (20 live fragments = 80 registers beside the 128 accumulators: with all 24 live hipcc spilled).  Late reads
                // of the SLAB are safe — nobody writes the current slice's slab: the partner group's LOAD phases that run beside this
                // group's MFMA phases fill the other slab, and only in the dx = 1 phase (beside this group's dx = 0 MFMAs of the same
                // slice); late reads of the weight ring would race with the partner's refill of the slot.
                dma_phase_next(std::integral_constant<int, (dx + 2) % 3>{}, ph + 2);
                if (dx == 1) {
                    dma_slab_piece(std::integral_constant<int, 0>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 1>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 2>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 3>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 4>{}, csn, slab_next);
                }
                bf16x8 a[3][4], b[4][2];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) a[dy][rb] = lds_read16(smem + (wa + SOFF + dy * BTAP + rb * 16 * 64));
#pragma unroll
                for (int hr = 0; hr < 4; ++hr)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) b[hr][hf] = lds_read16(smem + (pb[dx] + hr * (HP * 64) + hf * 16 * 64));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // everything requested before this phase's DMAs has landed: this wave's pieces of the weights of phase ph + 1 and (dx = 2) of
                // the next slice's slab (and, in a tile's first phase, the previous tile's stores)
                cvk_wait_vm<3 + (dx == 1 ? 5 : 0)>();
                phase_barrier();
                // ======== MFMA phase: 96 MFMAs = 1536 matrix cycles between two barriers.  b[k] holds halo row k, later row k + 4.
                __builtin_amdgcn_s_setprio(1);
                auto mm = [&](int dy, int tp, int slot) {           // the 8 MFMAs of tile row tp and tap dy; its halo row sits in b[slot]
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int rb = 0; rb < 4; ++rb)
                            acc[rb][2 * tp + hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy][rb], b[slot][hf], acc[rb][2 * tp + hf], 0, 0, 0);
                };
                mm(0, 0, 0);                                        // halo row 0: its last use
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) b[0][hf] = lds_read16(smem + (pb[dx] + 4 * (HP * 64) + hf * 16 * 64));      // halo row 4
                __builtin_amdgcn_sched_barrier(0);
                mm(0, 1, 1); mm(0, 2, 2); mm(0, 3, 3);
                mm(1, 0, 1);                                        // halo row 1: its last use
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) b[1][hf] = lds_read16(smem + (pb[dx] + 5 * (HP * 64) + hf * 16 * 64));      // halo row 5
                __builtin_amdgcn_sched_barrier(0);
                mm(1, 1, 2); mm(1, 2, 3); mm(1, 3, 0);
                mm(2, 0, 2); mm(2, 1, 3); mm(2, 2, 0); mm(2, 3, 1);
                if (dx == 2) {
#pragma unroll
                    for (int dx2 = 0; dx2 < 3; ++dx2) pb[dx2] += pb_flip;
                    pb_flip = -pb_flip;
                }
                __builtin_amdgcn_s_setprio(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the late reads were consumed above; nothing of this wave's is queued)
                phase_barrier();
                ++ph;
            };
            col_body(std::integral_constant<int, 0>{}); col_body(std::integral_constant<int, 1>{}); col_body(std::integral_constant<int, 2>{});
        }
        if (grp == 0) phase_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the tail re-loads have landed: ring and slabs are free
        phase_barrier();

        // this tile's bias values BEFORE the next tile's DMAs go out: hipcc waits vmcnt(0) for a register load, i.e. for every DMA
        // issued in front of its use as well (tools/tile_stamps_h.py: 2.3 us per tile = the next slab's HBM latency, the very thing the
        // early prologue is there to hide)
        const int n0 = cur.nt * BN;
        f32x4 bvals[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int co = n0 + wc * 64 + rb * 16 + 4 * q4;
            bvals[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && co < Cout) bvals[rb] = *reinterpret_cast<const f32x4*>(bias + co);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) asm volatile("" : "+v"(bvals[rb]));

        // ---- the next tile's prologue goes out before this tile's epilogue ---------------------------------------------------------
        base += G;
        const int n_k = min(G, ntiles - base);
        const bool has_next = (int)blockIdx.x < n_k;
        const int next = has_next ? base + cvk_xcd_remap(blockIdx.x, n_k) : 0;
        Geo nxt = cur;
        if (has_next) {
            nxt = geo_of(next);
            setup_dma(nxt);
            issue_prologue();
        }

        // ---- epilogue: bias, statistics, pack; transposed through the stage in two passes; 16-byte buffer stores -------------------
        // acc[rb][cb][i]: channel n0 + wc*64 + rb*16 + 4*q4 + i, pixel (y0 + row0 + (cb >> 1), x0 + (cb & 1)*16 + l15)
        // the epilogue's lane terms are recomputed per tile from a laundered lane id: hoisted out of the tile loop (they are loop
        // invariant) the ~60 stage / store addresses would live in registers through the K loop and spill
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int l15 = elane & 15, q4 = elane >> 4, lane = elane;

        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Y + ((size_t)(cur.img * H + cur.y0) * W + cur.x0) * ldy + n0), 0, 0x7FFFFFFF, 0x00020000);
        // Every wave: bias, statistics, pack to bf16 IN PLACE (the packed pair of a block replaces the first two of its four
        // accumulator registers: no second register array beside the 128 accumulators).
        // Statistics: a lane holds (sum, sum of squares) of 16 channels over its 8 pixel blocks; the 16 lanes of a DPP row hold the
        // same channels for 16 pixel columns: four DPP adds per value leave the row total in every lane; 4 partials per channel
        // (2 groups x 2 row halves) meet in LDS and are combined in fp64 in a fixed order.
        {
            float s[4][4], q[4][4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x4 bv = bvals[rb];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[rb][j] = 0.f; q[rb][j] = 0.f; }
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const bool ok = (cur.x0 + (cb & 1) * 16 + l15 < W) & (cur.y0 + row0 + (cb >> 1) < H);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[rb][cb][j] + bv[j];
                    if (STATS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float vm = ok ? v[j] : 0.f;
                            s[rb][j] += vm;
                            q[rb][j] += vm * vm;
                        }
                    }
                    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    const float2 of = __builtin_bit_cast(float2, o);
                    acc[rb][cb][0] = of.x;
                    acc[rb][cb][1] = of.y;
                    __builtin_amdgcn_sched_barrier(0);      // one block at a time
                }
            }
            if (STATS) {
                float2* const red = reinterpret_cast<float2*>(smem + RED_OFF);
                const int part = grp * 2 + wp;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ss = row_sum16(s[rb][j]), qq = row_sum16(q[rb][j]);
                        if (l15 == 0) red[(wc * 64 + rb * 16 + 4 * q4 + j) * 4 + part] = float2{ss, qq};
                    }
            }
        }
        auto stage_mine = [&]() {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const int p = (wp * 4 + (cb >> 1)) * 32 + (cb & 1) * 16 + l15, chunk = wc * 8 + rb * 2 + (q4 >> 1);
                    *reinterpret_cast<float2*>(smem + STAGE_OFF + p * 256 + ((chunk ^ (p & 15)) << 4) + 8 * (q4 & 1)) = float2{acc[rb][cb][0], acc[rb][cb][1]};
                }
        };
        auto store_pass = [&](int pass) {
            // wave w stores stage rows 32w .. 32w + 31 (= tile row 8*pass + w): 8 instructions of 4 pixels x 256 B
            const int chunk = lane & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = wave * 32 + it * 4 + (lane >> 4);
                const int prow = pass * 8 + (p >> 5), pcol = p & 31;
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + STAGE_OFF + p * 256 + ((chunk ^ (p & 15)) << 4));
                const bool ok = (cur.y0 + prow < H) & (cur.x0 + pcol < W) & (n0 + chunk * 8 < ldy);
                const unsigned off = (unsigned)((prow * W + pcol) * ldy + chunk * 8) * 2u;
                if (nts) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 0);
            }
        };
        if (grp == 0) stage_mine();
        __syncthreads();
        store_pass(0);
        __syncthreads();
        if (grp == 1) stage_mine();
        __syncthreads();
        store_pass(1);
        __syncthreads();
        if (STATS) {
            // every partial was written before the last barrier above
            if (tid < BN) {
                const float2* const red = reinterpret_cast<const float2*>(smem + RED_OFF);
                double S = 0.0, Q = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float2 v = red[tid * 4 + i]; S += (double)v.x; Q += (double)v.y; }
                const int co = n0 + tid;
                const int nvalid = min(TH, H - cur.y0) * min(TW, W - cur.x0);
                if (co < Cout) {
                    const double m2 = Q - S * S / (double)nvalid;
                    stats[(size_t)cur.sp * Cout + co] = (float)S;
                    stats[(size_t)(P + cur.sp) * Cout + co] = (float)(m2 > 0.0 ? m2 : 0.0);
                }
                if (cur.nt == 0 && tid == 0) cnt[cur.sp] = (float)nvalid;
            }
            // the partials are rewritten in the next tile's epilogue, at least four barriers from here
        }
        if (!has_next) break;
        tile = next;
        cur = nxt;
        stores_in_flight = true;
    }
}

#endif  // CVK_EXPERIMENTS (k_conv_bf16c)

// ---------------------------------------------------------------------------------------------- 64 output channels per workgroup
// Layers with <= 64 output channels and >= 64 input channels (ups4 / up4.0: 128 -> 64 at full resolution, 64 -> 64, and the data-grad
// of down2.0), which k_conv_bf16s<64> ran at 0.25-0.29 of the bf16 peak.  Same machine as k_conv_bf16q — persistent, two groups of four
// waves alternating LOAD and MFMA phases, 16 x 32 pixel tile, 16x16x32 MFMA, scalar-addressed DMA, transposed epilogue — with half
// the output channels per workgroup: a wave owns 64 channels x 64 pixels (2 tile rows), 64 accumulator registers.  One tap would be
// only 16 MFMAs (256 cycles) against a barrier per phase and 8 fragment reads, so a phase is a KERNEL ROW: the three taps dx = 0, 1, 2 of
// one dy — 48 MFMAs (768 cycles), 24 fragment reads (96 registers, affordable beside 64 accumulators), three phases per channel slice.
// A ring slot holds the three 4 KiB weight tiles of a kernel row (12 KiB, contiguous in the tile-major pack); slot = dy.  DMA per wave and
// phase: 1 weight piece (waves 0-3: 2) + 3 / 2 / 0 slab pieces (the next slice's slab must have landed before its first phase reads it).
// LDS: ring 36 KiB | slab A 40 | slab B 40 | (epilogue) stage 64 KiB over slab B and beyond: the next tile's prologue (slab A, ring slots
// 0 and 1) is issued before the epilogue, as in k_conv_bf16q, and the whole tile is staged in ONE pass (512 pixels x 128 B).
constexpr int HBN = 64;                          // output channels per workgroup
constexpr int HROW = 3 * HBN * 64;               // 12 KiB: the weight tiles of one kernel row of one channel slice

// P128: the weights come in the 128-row tile-major pack of k_conv_bf16q (layers with > 64 output channels whose tile count fills the chip
// badly with 128-channel tiles, e.g. 512 channels at 90x120: 384 tiles = 1.5 rounds of 256 CUs, 768 half-width tiles = 3): this workgroup's
// 64 rows are one half of every 8 KiB tap tile — three 4 KiB pieces 8 KiB apart instead of 12 contiguous KiB, still scalar-addressed.
template <bool STATS, int DBG = 0, bool P128 = false, bool PCOL = false, bool MST = true>      // MST: statistics on the matrix pipe (round 5)     // DBG 1 (instantiated in the experiments build only): s_memrealtime stamps of workgroup 0's first 16 tiles -> `stats`
__global__ __launch_bounds__(512, 2) void k_conv_bf16h(const __bf16* __restrict__ X, const char* __restrict__ Wp,
                                                      const float* __restrict__ bias, __bf16* __restrict__ Y,
                                                      float* __restrict__ stats, float* __restrict__ cnt, int H, int W, int Cin,
                                                      int Cout, int ldy, int tilesX, int tilesY, int tilesN, int P, int ntiles, int nts) {
    constexpr int RING_BYTES = 3 * HROW;                          // [0, 36 KiB) ring of three kernel rows; then slab A, slab B
    constexpr int STAGE_OFF = RING_BYTES + SLAB_BYTES;            // epilogue stage: 512 pixels x 128 B
    constexpr int STAGE_BYTES = TH * TW * HBN * 2;
    constexpr int RED_OFF = STAGE_OFF + STAGE_BYTES;              // statistics partials [channel 64][partial 8] (sum, sumsq)
    constexpr int LDS_BYTES = RED_OFF + (STATS ? HBN * 8 * 8 : 0);
    constexpr int NSTORE = 8;
    static_assert(LDS_BYTES >= RING_BYTES + 2 * SLAB_BYTES && LDS_BYTES <= 160 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const unsigned smem_addr = cvk_lds_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int grp = wave >> 2, wp = wave & 3;
    const int row0 = grp * 8 + wp * 2;         // this wave's two tile rows
    const bool lo = wave < 4;                  // waves 0-3 move two weight pieces per phase, waves 4-7 one
    const int ncs = Cin / CK, nph = ncs * 3;
    const int G = gridDim.x;

    struct Geo { int nt, sp, x0, y0, img; };
    const TileDiv divN(tilesN), divX(tilesX), divXY(tilesX * tilesY);
    auto geo_of = [&](int t) {
        Geo g;
        g.sp = divN.div(t);
        g.nt = divN.mod(t, g.sp);
        g.img = divXY.div(g.sp);
        const int rem = divXY.mod(g.sp, g.img);
        const int ty = divX.div(rem), tx = divX.mod(rem, ty);
        g.x0 = tx * TW; g.y0 = ty * TH;
        return g;
    };
    unsigned aoff[5];
    i32x4 xrsrc;
    xrsrc[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)(H * W * Cin) * 2u));
    xrsrc[3] = 0x00020000;
    constexpr int WROW = P128 ? 3 * BTAP : HROW;                 // source bytes of one kernel row of one slice
    // source offset of this wave's first piece inside a kernel row (piece p = tap p / 4, KiB p % 4 of the tap's 64 rows)
    const unsigned wvoff = (P128 ? (wave >> 2) * (PCOL ? 3 * BTAP : BTAP) + (wave & 3) * 1024 : wave * 1024) + lane * 16;      // PCOL: the pack is column-major (tap (dy, dx) at position 3 dx + dy)
    const unsigned wave_lds = smem_addr + wave * 1024;
    const char* wnext = Wp;
    int slab_yx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int row = (8 * t + wave) * 16 + (lane >> 2);
        const int hy = row / HP;
        slab_yx[t] = row < SLAB_ROWS ? (hy << 8) | (row - hy * HP) : 0x4000;
    }
    auto setup_dma = [&](const Geo& g) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int hy = slab_yx[t] >> 8, hx = slab_yx[t] & 255;
            const int chunk = (lane & 3) ^ (((hx >> 2) & 1) << 1);
            const int iy = g.y0 - 1 + hy, ix = g.x0 - 1 + hx;
            const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            aoff[t] = ok ? (unsigned)((iy * W + ix) * Cin + chunk * 8) * 2u : 0x80000000u;
        }
        const uintptr_t xbase = (uintptr_t)(X + (size_t)g.img * H * W * Cin);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xbase);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32) & 0xFFFF);
        wnext = P128 ? Wp + (size_t)(g.nt >> 1) * nph * WROW + (g.nt & 1) * (HBN * 64) : Wp + (size_t)g.nt * nph * WROW;
    };
    auto dma_slab_piece = [&](auto t_tag, int cs, unsigned slab_lds) {
        constexpr int T = decltype(t_tag)::value;
        dma16_buf_i<T * 8192>(aoff[T], xrsrc, (unsigned)cs * (CK * 2), slab_lds);
    };
    // the weights of the next kernel row (12 pieces) into ring slot SLOT: piece `wave` by every wave, piece 8 + wave by waves 0-3
    auto dma_row_next = [&](auto slot_tag, int ph) {
        constexpr int SLOT = decltype(slot_tag)::value;
        dma16_saddr_i<SLOT * HROW>(wvoff, wnext, wave_lds);
        if (lo) dma16_saddr_i<SLOT * HROW + 8192>(wvoff, wnext + (P128 ? (PCOL ? 6 * BTAP : 2 * BTAP) : 8192), wave_lds);      // pieces 8..11 = the third tap
        if (ph < nph - 1) wnext += (P128 && PCOL) ? (SLOT == 2 ? 7 * BTAP : BTAP) : WROW;      // column-major pack: kernel row dy starts at position dy, the next slice 9 tiles on
    };
    auto issue_prologue = [&]() {
        dma_slab_piece(std::integral_constant<int, 0>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 1>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 2>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 3>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 4>{}, 0, wave_lds + RING_BYTES);
        dma_row_next(std::integral_constant<int, 0>{}, 0);
    };

    // weights: tap dx of the slot's kernel row at dx * 4 KiB, row rb*16 + l15, chunk q4; pixels: halo row row0 + tp + dy, halo column
    // half*16 + l15 + dx, chunk q4
    const int wa = l15 * 64 + ((q4 ^ (((l15 >> 2) & 1) << 1)) << 4);
    int pb0[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) pb0[dx] = RING_BYTES + row0 * (HP * 64) + (l15 + dx) * 64 + ((q4 ^ ((((l15 + dx) >> 2) & 1) << 1)) << 4);

    int base = 0;
    int tile = cvk_xcd_remap(blockIdx.x, min(G, ntiles));
    Geo cur = geo_of(tile);
    setup_dma(cur);
    issue_prologue();
    bool stores_in_flight = false;
    int tcount = 0;
    auto stamp = [&](int which) {
        if (DBG && blockIdx.x == 0 && tcount < 16) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            // STATS builds keep their statistics: the stamps go behind the counts (the timing script allocates the room)
            unsigned long long* const sb = STATS ? reinterpret_cast<unsigned long long*>(cnt + ((P + 1) & ~1)) : reinterpret_cast<unsigned long long*>(stats);
            if (tid == 0) sb[tcount * 8 + which] = t;
        }
    };

    while (true) {
        stamp(0);
        // kernel row 1 is requested HERE, behind the previous tile's stores (round 5, as in k_conv_bf16q: no special case for a tile's first
        // phase in the K loop, no peeled first slice)
        dma_row_next(std::integral_constant<int, 1>{}, 1);
        // slab of slice 0 and the weights of kernel row 0 have landed; behind them: (after the first tile) the previous tile's NSTORE
        // stores and the row just requested (2 / 1 pieces)
        if (lo) { if (stores_in_flight) cvk_wait_vm<2 + NSTORE>(); else cvk_wait_vm<2>(); }
        else    { if (stores_in_flight) cvk_wait_vm<1 + NSTORE>(); else cvk_wait_vm<1>(); }
        phase_barrier();
        stamp(1);
        if (grp == 1) phase_barrier();

        f32x4v acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
        int pb[3] = {pb0[0], pb0[1], pb0[2]};
        int pb_flip = SLAB_BYTES;

        int ph = 0;
        for (int cs = 0; cs < ncs; ++cs) {
            const unsigned slab_next = wave_lds + RING_BYTES + ((cs + 1) & 1) * SLAB_BYTES;
            const int csn = min(cs + 1, ncs - 1);
            auto row_body = [&](auto dy_tag) {
                constexpr int dy = decltype(dy_tag)::value;
                // ======== LOAD phase: weights of kernel row ph + 2, the next slice's slab (3 + 2 pieces), this row's 24 fragments
                dma_row_next(std::integral_constant<int, (dy + 2) % 3>{}, ph + 2);
                if (dy == 0) {
                    dma_slab_piece(std::integral_constant<int, 0>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 1>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 2>{}, csn, slab_next);
                } else if (dy == 1) {
                    dma_slab_piece(std::integral_constant<int, 3>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 4>{}, csn, slab_next);
                }
                bf16x8 a[3][4], b[3][4];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) a[dx][rb] = lds_read16(smem + (wa + dy * HROW + dx * (HBN * 64) + rb * 16 * 64));
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) b[dx][cb] = lds_read16(smem + (pb[dx] + ((cb >> 1) + dy) * (HP * 64) + (cb & 1) * 16 * 64));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // the weights of kernel row ph + 1 have landed: everything requested before this phase's DMAs (2 or 1 weight pieces + 3 / 2 / 0
                // slab pieces) — except in the first phase of a later tile, whose row-1 weights sit in front of the previous tile's stores
                {
                    constexpr int NS = dy == 0 ? 3 : (dy == 1 ? 2 : 0);
                    if (lo) cvk_wait_vm<2 + NS>(); else cvk_wait_vm<1 + NS>();
                }
                phase_barrier();
                // ======== MFMA phase: 48 MFMAs
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dx][rb], b[dx][cb], acc[rb][cb], 0, 0, 0);
                if (dy == 2) {
#pragma unroll
                    for (int dx2 = 0; dx2 < 3; ++dx2) pb[dx2] += pb_flip;
                    pb_flip = -pb_flip;
                }
                __builtin_amdgcn_s_setprio(0);
                phase_barrier();
                ++ph;
            };
            row_body(std::integral_constant<int, 0>{}); row_body(std::integral_constant<int, 1>{}); row_body(std::integral_constant<int, 2>{});
        }
        if (grp == 0) phase_barrier();
        stamp(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        phase_barrier();
        stamp(3);

        // this tile's bias values BEFORE the next tile's DMAs go out: hipcc waits vmcnt(0) for a register load, i.e. for every DMA
        // issued in front of its use as well (measured: 2.3 us per tile with the loads inside the epilogue = the slab's HBM latency)
        const int n0 = cur.nt * HBN;
        f32x4 bvals[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int co = n0 + rb * 16 + 4 * q4;
            bvals[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && co < Cout) bvals[rb] = *reinterpret_cast<const f32x4*>(bias + co);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) asm volatile("" : "+v"(bvals[rb]));      // the loads stay in front of the DMAs below

        base += G;
        const int n_k = min(G, ntiles - base);
        const bool has_next = (int)blockIdx.x < n_k;
        const int next = has_next ? base + cvk_xcd_remap(blockIdx.x, n_k) : 0;
        Geo nxt = cur;
        if (has_next) {
            nxt = geo_of(next);
            setup_dma(nxt);
        }
        stamp(4);
        // The next tile's prologue (7-9 KiB per wave, 64 KiB per workgroup: ~1000 cycles of the CU's 64 B/clk LDS-DMA path) goes out one
        // piece per block of the epilogue below instead of in one burst in front of it: the waves compute while the queue drains
        // (tools/tile_stamps_h.py: "next prologue issue" 1.2-1.5 us of a 10-16 us tile).  All of it is still issued before the first store.
        auto prologue_piece = [&](int k) {
            if (!has_next) return;
            if (k == 0) dma_slab_piece(std::integral_constant<int, 0>{}, 0, wave_lds + RING_BYTES);
            if (k == 1) dma_slab_piece(std::integral_constant<int, 1>{}, 0, wave_lds + RING_BYTES);
            if (k == 2) dma_slab_piece(std::integral_constant<int, 2>{}, 0, wave_lds + RING_BYTES);
            if (k == 3) dma_slab_piece(std::integral_constant<int, 3>{}, 0, wave_lds + RING_BYTES);
            if (k == 4) dma_slab_piece(std::integral_constant<int, 4>{}, 0, wave_lds + RING_BYTES);
            if (k == 5) dma_row_next(std::integral_constant<int, 0>{}, 0);
        };

        // ---- epilogue: acc[rb][cb][i] = channel n0 + rb*16 + 4*q4 + i, pixel (y0 + row0 + (cb >> 1), x0 + (cb & 1)*16 + l15) ---------------
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int l15 = elane & 15, q4 = elane >> 4, lane = elane;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Y + ((size_t)(cur.img * H + cur.y0) * W + cur.x0) * ldy + n0), 0, 0x7FFFFFFF, 0x00020000);
        // Statistics (round 5, MST): NOT from the accumulators with vector instructions (16 values x (mask, add, multiply-add) + 32 row sums of four
        // DPP steps each: ~640 of the 1440 vector instructions of this epilogue, 1.5 us of a 9-16 us tile with both waves of a SIMD in it) but on
        // the matrix pipe, which is idle here, from the STAGED tile — the bf16 values BatchNorm will normalise (oracle/bf16_emul.py takes its
        // statistics from the rounded tensor as well).  A wave reads its 64 pixels x 64 channels back transposed (ds_read_b64_tr_b16: lane =
        // channel, 8 consecutive pixels = the k of v_mfma_f32_16x16x32_bf16) and issues per 16-channel block  S = A x ones  and  Q = A x A^T
        // (the operand registers serve as B too): every column of S holds the channel sums, the diagonal of Q the sums of squares — 16 MFMAs
        // and 16 reads per wave.  Out-of-frame pixels of a ragged tile are staged as zeros.
        const bool ragged = (cur.y0 + TH > H) | (cur.x0 + TW > W);
        {
            float s[4][4], q[4][4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x4 bv = bvals[rb];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[rb][j] = 0.f; q[rb][j] = 0.f; }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const bool ok = (cur.x0 + (cb & 1) * 16 + l15 < W) & (cur.y0 + row0 + (cb >> 1) < H);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[rb][cb][j] + bv[j];
                    if (STATS && !MST) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float vm = ok ? v[j] : 0.f;
                            s[rb][j] += vm;
                            q[rb][j] += vm * vm;
                        }
                    }
                    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    if (STATS && MST && ragged && !ok) o = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    // stage: [512 pixels][128 B], 16-byte chunk c of pixel p at position c ^ ((p >> 1) & 7)
                    const int p = (row0 + (cb >> 1)) * 32 + (cb & 1) * 16 + l15, chunk = rb * 2 + (q4 >> 1);
                    *reinterpret_cast<bf16x4*>(smem + STAGE_OFF + p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4) + 8 * (q4 & 1)) = o;
                    if (rb * 4 + cb < 6) prologue_piece(rb * 4 + cb);
                }
            }
            if (STATS && !MST) {
                float2* const red = reinterpret_cast<float2*>(smem + RED_OFF);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ss = row_sum16(s[rb][j]), qq = row_sum16(q[rb][j]);
                        if (l15 == 0) red[(rb * 16 + 4 * q4 + j) * 8 + wave] = float2{ss, qq};
                    }
            }
        }
        __syncthreads();
        stamp(5);
        {
            // wave w stores tile rows 2w, 2w + 1: 8 instructions of 8 pixels x 128 B
            const int chunk = lane & 7;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = wave * 64 + it * 8 + (lane >> 3);
                const int prow = p >> 5, pcol = p & 31;
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + STAGE_OFF + p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4));
                const bool ok = (cur.y0 + prow < H) & (cur.x0 + pcol < W) & (n0 + chunk * 8 < ldy);
                const unsigned off = (unsigned)((prow * W + pcol) * ldy + chunk * 8) * 2u;
                if (nts) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 0);
            }
        }
        if (STATS && MST) {
            // wave w: pixels 64 w .. 64 w + 63 (the rows it just stored).  Lane (l15, q4) of a 16-lane group addresses 4 consecutive channels
            // (4 (l15 & 3)) of pixel row 8 q4 + (l15 >> 2) (+ 4 for the second read); the transposing read hands lane l15 channel l15 of
            // the group's four rows.
            typedef short s16x4t __attribute__((ext_vector_type(4)));
            const int tq = l15 >> 2, tp = l15 & 3;
            int ra[2];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const int pl = 8 * q4 + tq + 4 * r2;                      // pixel inside a 32-pixel block: (p >> 1) & 7 only depends on it
                ra[r2] = STAGE_OFF + (wave * 64 + pl) * 128 + (tp & 1) * 8;
            }
            const int swz[2] = {((8 * q4 + tq) >> 1) & 7, ((8 * q4 + tq + 4) >> 1) & 7};
            bf16x8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.f;
            f32x4v aS[4], aQ[4];
#pragma unroll
            for (int cbk = 0; cbk < 4; ++cbk) { aS[cbk] = f32x4v{0.f, 0.f, 0.f, 0.f}; aQ[cbk] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int cbk = 0; cbk < 4; ++cbk) {
                    const int chunk = cbk * 2 + (tp >> 1);
                    const s16x4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4t*)(smem + ra[0] + pb * 32 * 128 + ((chunk ^ swz[0]) << 4)));
                    const s16x4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4t*)(smem + ra[1] + pb * 32 * 128 + ((chunk ^ swz[1]) << 4)));
                    const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    aS[cbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, aS[cbk], 0, 0, 0);
                    aQ[cbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, aQ[cbk], 0, 0, 0);
                }
            float2* const red = reinterpret_cast<float2*>(smem + RED_OFF);       // [channel 64][wave 8] (sum, sum of squares)
            // aS[cbk][j]: channel 16 cbk + 4 q4 + j, identical in the 16 lanes of a group; aQ[cbk][j] in lane l15: (row 4 q4 + j, column l15):
            // the diagonal sits in the lanes with l15 >> 2 == q4, register l15 & 3
#pragma unroll
            for (int cbk = 0; cbk < 4; ++cbk) {
                const float qd = tp == 0 ? aQ[cbk][0] : (tp == 1 ? aQ[cbk][1] : (tp == 2 ? aQ[cbk][2] : aQ[cbk][3]));
                const float sd = tp == 0 ? aS[cbk][0] : (tp == 1 ? aS[cbk][1] : (tp == 2 ? aS[cbk][2] : aS[cbk][3]));
                if (tq == q4) red[(cbk * 16 + l15) * 8 + wave] = float2{sd, qd};
            }
            __syncthreads();
        }
        if (STATS) {
            if (tid < HBN) {
                const float2* const red = reinterpret_cast<const float2*>(smem + RED_OFF);
                double S = 0.0, Q = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float2 v = red[tid * 8 + i]; S += (double)v.x; Q += (double)v.y; }
                const int co = n0 + tid;
                const int nvalid = min(TH, H - cur.y0) * min(TW, W - cur.x0);
                if (co < Cout) {
                    // 1 / n through the fp32 reciprocal (exact for the 512 pixels of a whole tile): an fp64 division per thread was ~40
                    // instructions of one wave with seven waiting at the barrier below
                    const double m2 = Q - S * S * (double)(1.0f / (float)nvalid);
                    stats[(size_t)cur.sp * Cout + co] = (float)S;
                    stats[(size_t)(P + cur.sp) * Cout + co] = (float)(m2 > 0.0 ? m2 : 0.0);
                }
                if (cur.nt == 0 && tid == 0) cnt[cur.sp] = (float)nvalid;
            }
        }
        __syncthreads();            // stage and partials are read: the next tile's K loop may overwrite slab B
        stamp(6);
        ++tcount;
        if (!has_next) break;
        tile = next;
        cur = nxt;
        stores_in_flight = true;
    }
}

// EXPERIMENTS BUILD ONLY (CVK_BF16H_PIPE=1): parity green (tests/test_gpu_bf16.py), but interleaved on one box the four 64-output-channel
// forward launches and their data-grads run 3-5 % SLOWER than k_conv_bf16h (down1.1 257-264 -> 265-273 us, ups4 370-373 -> 387-393 us):
// hiding the 4.4 us a tile spends outside its 5.4 us K loop bought nothing, like the kernel-column phases before it — on this part these
// kernels' time follows the work done per tile (matrix + vector + LDS instructions under a power cap), not the bubbles between them; what
// did pay is REMOVING work (the matrix-pipe statistics: -3.5 %).  DESIGN.md §5b round 5.
#ifdef CVK_EXPERIMENTS
// ---------------------------------------------------------------------------------------------- 64 output channels, PIPELINED EPILOGUE (round 5)
// k_conv_bf16h with a tile's epilogue moved under the NEXT tile's K loop.  With 64 (128) input channels the K loop of a 64-channel tile is
// 5.4 (11) us and everything outside it 4.4-5.7 us — bias, pack, transposed store, statistics, with both wave groups in it together and the
// matrix pipe idle (tools/tile_stamps_h.py).  The microbenchmark of round 5 says vector / LDS work of one wave costs its SIMD partner's
// MFMA stream nothing.  So:
//   * at the end of its K loop a wave only adds the bias and packs its 64 accumulators to bf16 IN REGISTERS (32 VGPRs, kept across the
//     tile boundary), the next tile's prologue goes out and the next K loop starts at once;
//   * in the LOAD phase of the next tile's kernel row 1 — beside the partner group's MFMA phase — the wave writes the packed tile into its
//     8 KiB of a 32 KiB stage, reads it back transposed (rows of 128 contiguous bytes per pixel), issues the 16-byte streaming stores and
//     (STATS) the matrix-pipe statistics of csrc note "MST".  The stage is a per-WAVE transpose (a wave stores the two tile rows it
//     computed), so nothing but the wave's own LDS order is needed; waves w and w + 4 share a region because their LOAD phases never
//     coincide and each finishes all LDS work on it inside one phase;
//   * the 8 x 2 statistics partials meet in LDS and are finalised by wave 0 in ITS next LOAD phase (kernel row 2), one barrier pair after
//     the partner group wrote its half;
//   * the stores are the newest entries of the in-order counter in that phase (counted vmcnt + 8) and old ones a phase later.
// After its last tile a workgroup runs the same pieces once without a K loop around them.  And the phase sequence never stops at a tile
// boundary: the next tile's first slab and kernel rows are requested in the current tile's last slice, in the slots k_conv_bf16h filled with
// dead re-loads (a first version that issued the prologue at the boundary was 6-10 % SLOWER than k_conv_bf16h: with the epilogue gone
// nothing covered the first slab's HBM latency).  Needs an even number of channel slices (Cin % 64 == 0) and Cout <= 1024 (bias in LDS).
// LDS: ring 36 | slab A 40 | slab B 40 | stage 32 | partials 4 | bias 4 = 156 KiB.
template <bool STATS, bool P128>
__global__ __launch_bounds__(512, 2) void k_conv_bf16hp(const __bf16* __restrict__ X, const char* __restrict__ Wp,
                                                       const float* __restrict__ bias, __bf16* __restrict__ Y,
                                                       float* __restrict__ stats, float* __restrict__ cnt, int H, int W, int Cin,
                                                       int Cout, int ldy, int tilesX, int tilesY, int tilesN, int P, int ntiles, int nts) {
    constexpr int RING_BYTES = 3 * HROW;
    constexpr int STAGE_OFF = RING_BYTES + 2 * SLAB_BYTES;        // four wave regions of 64 pixels x 128 B
    constexpr int STAGE_BYTES = 4 * 64 * HBN * 2;
    constexpr int RED_OFF = STAGE_OFF + STAGE_BYTES;              // statistics partials [channel 64][wave 8] (sum, sumsq)
    constexpr int BIAS_OFF = RED_OFF + HBN * 8 * 8;               // the layer's bias (<= 1024 channels), copied once per workgroup
    constexpr int LDS_BYTES = BIAS_OFF + 1024 * 4;
    constexpr int NSTORE = 8;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS plan");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const unsigned smem_addr = cvk_lds_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int grp = wave >> 2, wp = wave & 3;
    const int row0 = grp * 8 + wp * 2;
    const bool lo = wave < 4;
    const int ncs = Cin / CK, nph = ncs * 3;
    const int G = gridDim.x;

    struct Geo { int nt, sp, x0, y0, img; };
    const TileDiv divN(tilesN), divX(tilesX), divXY(tilesX * tilesY);
    auto geo_of = [&](int t) {
        Geo g;
        g.sp = divN.div(t);
        g.nt = divN.mod(t, g.sp);
        g.img = divXY.div(g.sp);
        const int rem = divXY.mod(g.sp, g.img);
        const int ty = divX.div(rem), tx = divX.mod(rem, ty);
        g.x0 = tx * TW; g.y0 = ty * TH;
        return g;
    };
    unsigned aoff[5];
    i32x4 xrsrc;
    xrsrc[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)(H * W * Cin) * 2u));
    xrsrc[3] = 0x00020000;
    constexpr int WROW = P128 ? 3 * BTAP : HROW;
    const unsigned wvoff = (P128 ? (wave >> 2) * BTAP + (wave & 3) * 1024 : wave * 1024) + lane * 16;
    const unsigned wave_lds = smem_addr + wave * 1024;
    const char* wnext = Wp;
    int slab_yx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int row = (8 * t + wave) * 16 + (lane >> 2);
        const int hy = row / HP;
        slab_yx[t] = row < SLAB_ROWS ? (hy << 8) | (row - hy * HP) : 0x4000;
    }
    auto setup_dma = [&](const Geo& g) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int hy = slab_yx[t] >> 8, hx = slab_yx[t] & 255;
            const int chunk = (lane & 3) ^ (((hx >> 2) & 1) << 1);
            const int iy = g.y0 - 1 + hy, ix = g.x0 - 1 + hx;
            const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            aoff[t] = ok ? (unsigned)((iy * W + ix) * Cin + chunk * 8) * 2u : 0x80000000u;
        }
        const uintptr_t xbase = (uintptr_t)(X + (size_t)g.img * H * W * Cin);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xbase);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32) & 0xFFFF);
        wnext = P128 ? Wp + (size_t)(g.nt >> 1) * nph * WROW + (g.nt & 1) * (HBN * 64) : Wp + (size_t)g.nt * nph * WROW;
    };
    auto dma_slab_piece = [&](auto t_tag, int cs, unsigned slab_lds) {
        constexpr int T = decltype(t_tag)::value;
        dma16_buf_i<T * 8192>(aoff[T], xrsrc, (unsigned)cs * (CK * 2), slab_lds);
    };
    auto dma_row_next = [&](auto slot_tag, bool advance) {      // the kernel row wnext points at into ring slot SLOT; then (advance) on to the next row
        constexpr int SLOT = decltype(slot_tag)::value;
        dma16_saddr_i<SLOT * HROW>(wvoff, wnext, wave_lds);
        if (lo) dma16_saddr_i<SLOT * HROW + 8192>(wvoff, wnext + (P128 ? 2 * BTAP : 8192), wave_lds);
        if (advance) wnext += WROW;
    };
    auto issue_prologue = [&]() {
        dma_slab_piece(std::integral_constant<int, 0>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 1>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 2>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 3>{}, 0, wave_lds + RING_BYTES);
        dma_slab_piece(std::integral_constant<int, 4>{}, 0, wave_lds + RING_BYTES);
        dma_row_next(std::integral_constant<int, 0>{}, true);
        dma_row_next(std::integral_constant<int, 1>{}, true);
    };
    const int wa = l15 * 64 + ((q4 ^ (((l15 >> 2) & 1) << 1)) << 4);
    int pb0[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) pb0[dx] = RING_BYTES + row0 * (HP * 64) + (l15 + dx) * 64 + ((q4 ^ ((((l15 + dx) >> 2) & 1) << 1)) << 4);

    // ---- the previous tile, packed: pk[rb][cb] = channels n0 + rb*16 + 4 q4 .. + 3 of pixel (row0 + (cb >> 1), (cb & 1)*16 + l15) as 4 bf16 ------
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    typedef short s16x4t __attribute__((ext_vector_type(4)));
    u32x2v pk[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) pk[a][b] = u32x2v{0u, 0u};
    Geo prev = geo_of(0);
    bool have_prev = false;
    const int wreg = STAGE_OFF + (wave & 3) * (64 * HBN * 2);      // this wave's stage region (shared with wave ^ 4: never in the same phase)

    // stage write, statistics partials, read-back + stores of the PREVIOUS tile: everything that touches the stage, inside one phase
    auto epi_lds_and_store = [&]() {
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int l15 = elane & 15, q4 = elane >> 4, lane = elane;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int p = (cb >> 1) * 32 + (cb & 1) * 16 + l15, chunk = rb * 2 + (q4 >> 1);
                *reinterpret_cast<u32x2v*>(smem + wreg + p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4) + 8 * (q4 & 1)) = pk[rb][cb];
            }
        if (STATS) {
            const int tq = l15 >> 2, tp = l15 & 3;
            const int ra0 = wreg + (8 * q4 + tq) * 128 + (tp & 1) * 8, ra1 = ra0 + 4 * 128;
            const int sw0 = ((8 * q4 + tq) >> 1) & 7, sw1 = ((8 * q4 + tq + 4) >> 1) & 7;
            bf16x8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.f;
            f32x4v aS[4], aQ[4];
#pragma unroll
            for (int cbk = 0; cbk < 4; ++cbk) { aS[cbk] = f32x4v{0.f, 0.f, 0.f, 0.f}; aQ[cbk] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int pbk = 0; pbk < 2; ++pbk)
#pragma unroll
                for (int cbk = 0; cbk < 4; ++cbk) {
                    const int chunk = cbk * 2 + (tp >> 1);
                    const s16x4t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4t*)(smem + ra0 + pbk * 32 * 128 + ((chunk ^ sw0) << 4)));
                    const s16x4t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4t*)(smem + ra1 + pbk * 32 * 128 + ((chunk ^ sw1) << 4)));
                    const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                    aS[cbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, aS[cbk], 0, 0, 0);
                    aQ[cbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, aQ[cbk], 0, 0, 0);
                }
            float2* const red = reinterpret_cast<float2*>(smem + RED_OFF);
#pragma unroll
            for (int cbk = 0; cbk < 4; ++cbk) {
                const float qd = tp == 0 ? aQ[cbk][0] : (tp == 1 ? aQ[cbk][1] : (tp == 2 ? aQ[cbk][2] : aQ[cbk][3]));
                const float sd = tp == 0 ? aS[cbk][0] : (tp == 1 ? aS[cbk][1] : (tp == 2 ? aS[cbk][2] : aS[cbk][3]));
                if (tq == q4) red[(cbk * 16 + l15) * 8 + wave] = float2{sd, qd};
            }
        }
        const int n0 = prev.nt * HBN;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Y + ((size_t)(prev.img * H + prev.y0) * W + prev.x0) * ldy + n0), 0, 0x7FFFFFFF, 0x00020000);
        const int chunk = lane & 7;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int p = it * 8 + (lane >> 3);
            const int prow = row0 + (p >> 5), pcol = p & 31;
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + wreg + p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4));
            const bool ok = (prev.y0 + prow < H) & (prev.x0 + pcol < W) & (n0 + chunk * 8 < ldy);
            const unsigned off = (unsigned)((prow * W + pcol) * ldy + chunk * 8) * 2u;
            if (nts) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrsrc, ok ? off : 0x80000000u, 0, 0);
        }
    };
    // wave 0: the eight partials of every channel in a fixed order, fp64; EXACTLY three always-issued stores (sum, M2, count: lanes
    // that have nothing to store aim past the buffer) so that the phase's counted vmcnt stays exact
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stats, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc((void*)cnt, 0, 0x7FFFFFFF, 0x00020000);
    auto epi_finalize = [&]() {
        if (wave == 0) {
            const float2* const red = reinterpret_cast<const float2*>(smem + RED_OFF);
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float2 v = red[lane * 8 + i]; S += (double)v.x; Q += (double)v.y; }
            const int co = prev.nt * HBN + lane;
            const int nvalid = min(TH, H - prev.y0) * min(TW, W - prev.x0);
            const double m2 = Q - S * S * (double)(1.0f / (float)nvalid);
            const bool cok = co < Cout;
            const unsigned o1 = cok ? (unsigned)(((size_t)prev.sp * Cout + co) * 4) : 0x80000000u;
            const unsigned o2 = cok ? (unsigned)(((size_t)(P + prev.sp) * Cout + co) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)S), srsrc, o1, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(m2 > 0.0 ? m2 : 0.0)), srsrc, o2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)nvalid), crsrc,
                                                  (prev.nt == 0 && lane == 0) ? (unsigned)prev.sp * 4u : 0x80000000u, 0, 0);
        }
    };

    // the layer's bias into LDS once (a register load later would wait for every LDS-DMA in flight: hipcc's vmcnt(0))
    for (int i = tid; i < 1024; i += 512) reinterpret_cast<float*>(smem + BIAS_OFF)[i] = (bias != nullptr && i < Cout) ? bias[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int base = 0;
    int tile = cvk_xcd_remap(blockIdx.x, min(G, ntiles));
    Geo cur = geo_of(tile);
    setup_dma(cur);
    issue_prologue();
    if (lo) cvk_wait_vm<2>(); else cvk_wait_vm<1>();        // slab of slice 0 and kernel row 0 have landed; row 1 (2 / 1 pieces) is behind them
    phase_barrier();
    if (grp == 1) phase_barrier();                          // group B runs one interval behind group A — for the whole kernel

    f32x4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    int pb[3] = {pb0[0], pb0[1], pb0[2]};
    int pb_flip = SLAB_BYTES;

    // ONE phase sequence over all tiles of the workgroup: the weight ring (three kernel rows, nph % 3 == 0) and the two slabs (ncs even)
    // run on across tile boundaries — the last slice of a tile requests the NEXT tile's first slab and its kernel rows 0 and 1 where
    // k_conv_bf16h re-loaded the last step into buffers nobody reads — so there is no prologue to wait for, no drain and no barrier
    // at a boundary; what is left there (bias, pack, 200-300 vector instructions) sits in the first LOAD phase's slack.
    while (true) {
        bool has_next = false;
        Geo nxt = cur;
        for (int cs = 0; cs < ncs; ++cs) {
            const bool lastcs = cs == ncs - 1;
            const unsigned slab_next = wave_lds + RING_BYTES + ((cs + 1) & 1) * SLAB_BYTES;
            auto row_body = [&](auto dy_tag) {
                constexpr int dy = decltype(dy_tag)::value;
                // ======== LOAD phase: the kernel row two phases ahead (in a tile's last slice: its last row, then rows 0 and 1 of the
                // next tile), the next slab (3 + 2 pieces; last slice: slice 0 of the next tile)
                if (!lastcs) dma_row_next(std::integral_constant<int, (dy + 2) % 3>{}, true);
                else if (dy == 0) {
                    dma_row_next(std::integral_constant<int, 2>{}, false);        // this tile's last kernel row
                    base += G;
                    const int n_k = min(G, ntiles - base);
                    has_next = (int)blockIdx.x < n_k;
                    if (has_next) {
                        nxt = geo_of(base + cvk_xcd_remap(blockIdx.x, n_k));
                        setup_dma(nxt);                                            // slab offsets / resource / weight stream of the next tile
                    }
                } else dma_row_next(std::integral_constant<int, (dy + 2) % 3>{}, has_next);   // no next tile: the last row again, into a dead slot
                const int csn = lastcs ? (has_next ? 0 : ncs - 1) : cs + 1;
                if (dy == 0) {
                    dma_slab_piece(std::integral_constant<int, 0>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 1>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 2>{}, csn, slab_next);
                } else if (dy == 1) {
                    dma_slab_piece(std::integral_constant<int, 3>{}, csn, slab_next);
                    dma_slab_piece(std::integral_constant<int, 4>{}, csn, slab_next);
                }
                // the previous tile's epilogue, in this tile's first slice: kernel row 1 = stage / statistics / stores, kernel row 2 =
                // the statistics' finalisation (wave 0; the partner group wrote its partials one interval ago)
                const bool epi = have_prev && cs == 0;
                if (dy == 1 && epi) { __builtin_amdgcn_sched_barrier(0); epi_lds_and_store(); __builtin_amdgcn_sched_barrier(0); }
                if (STATS && dy == 2 && epi) { __builtin_amdgcn_sched_barrier(0); epi_finalize(); __builtin_amdgcn_sched_barrier(0); }
                bf16x8 a[3][4], b[3][4];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) a[dx][rb] = lds_read16(smem + (wa + dy * HROW + dx * (HBN * 64) + rb * 16 * 64));
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) b[dx][cb] = lds_read16(smem + (pb[dx] + ((cb >> 1) + dy) * (HP * 64) + (cb & 1) * 16 * 64));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                {
                    // everything requested before this phase's DMAs has landed (kernel row ph + 1, the slab pieces of earlier phases); this
                    // phase's own requests — and the stores the epilogue pieces put behind them — may be in flight
                    constexpr int NS = dy == 0 ? 3 : (dy == 1 ? 2 : 0);
                    const bool st = (dy == 1) && epi;
                    const bool fz = STATS && (dy == 2) && epi && wave == 0;
                    if (lo) { if (st) cvk_wait_vm<2 + NS + NSTORE>(); else if (fz) cvk_wait_vm<2 + NS + 3>(); else cvk_wait_vm<2 + NS>(); }
                    else    { if (st) cvk_wait_vm<1 + NS + NSTORE>(); else cvk_wait_vm<1 + NS>(); }
                }
                phase_barrier();
                // ======== MFMA phase: 48 MFMAs
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dx][rb], b[dx][cb], acc[rb][cb], 0, 0, 0);
                if (dy == 2) {
#pragma unroll
                    for (int dx2 = 0; dx2 < 3; ++dx2) pb[dx2] += pb_flip;
                    pb_flip = -pb_flip;
                }
                __builtin_amdgcn_s_setprio(0);
                phase_barrier();
            };
            row_body(std::integral_constant<int, 0>{}); row_body(std::integral_constant<int, 1>{}); row_body(std::integral_constant<int, 2>{});
        }
        // ---- tile boundary: bias (from LDS), pack to bf16 in registers, clear the accumulators; out-of-frame pixels of a ragged tile as
        // zeros (the statistics read the staged tile)
        {
            const int n0 = cur.nt * HBN;
            const bool ragged = (cur.y0 + TH > H) | (cur.x0 + TW > W);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (n0 + rb * 16 + 4 * q4) * 4);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const bool ok = (cur.x0 + (cb & 1) * 16 + l15 < W) & (cur.y0 + row0 + (cb >> 1) < H);
                    bf16x4 o = {(__bf16)(acc[rb][cb][0] + bv[0]), (__bf16)(acc[rb][cb][1] + bv[1]),
                                (__bf16)(acc[rb][cb][2] + bv[2]), (__bf16)(acc[rb][cb][3] + bv[3])};
                    if (STATS && ragged && !ok) o = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    pk[rb][cb] = __builtin_bit_cast(u32x2v, o);
                    acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        prev = cur;
        have_prev = true;
        if (!has_next) break;
        cur = nxt;
    }
    if (grp == 0) phase_barrier();      // pairs with group B's last MFMA-phase barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the dead re-loads of the last tile have landed
    // the last tile's epilogue, without a K loop around it: group A's waves, then group B's (they share the stage regions)
    if (grp == 0) epi_lds_and_store();
    __syncthreads();
    if (grp == 1) epi_lds_and_store();
    __syncthreads();
    if (STATS) epi_finalize();
}

#endif  // CVK_EXPERIMENTS (k_conv_bf16hp)

// fp32 master weights, physical [Cout][3][3][Cin] -> tile-major bf16 pack [row tile][slice][tap][128 rows][4 chunks][8], chunk
// position p of row n holds source chunk p ^ ((n>>2)&3) (the LDS image of one DMA'd tap tile, byte for byte); zero padded.
// dgrad: rows are the INPUT channels of the layer, k runs over its output channels, taps rotated by 180 degrees.
__global__ void k_pack_w_pp(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int ntile, int ncs, int dgrad, int mf16, int bn, int col) {
    const size_t total = (size_t)ntile * ncs * 9 * bn * CK;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), p = (int)((i >> 3) & 3), n = (int)((i >> 5) & (bn - 1));
        size_t rest = (i >> 5) / bn;
        const int tpos = (int)(rest % 9); rest /= 9;
        const int tap = col ? (tpos % 3) * 3 + tpos / 3 : tpos;       // column-major stream (k_conv_bf16c): position 3 dx + dy holds tap (dy, dx)
        const int cs = (int)(rest % ncs);
        const int ntl = (int)(rest / ncs);
        const int row = ntl * bn + n;
        const int k = cs * CK + ((p ^ (mf16 ? ((n >> 2) & 1) << 1 : (n >> 2) & 3)) << 3) + e;
        float v = 0.f;
        if (!dgrad) { if (row < Cout && k < Cin) v = w[((size_t)row * 9 + tap) * Cin + k]; }
        else        { if (row < Cin && k < Cout) v = w[((size_t)k * 9 + (8 - tap)) * Cin + row]; }
        out[i] = (__bf16)v;
    }
}

// All weight packs of a step in ONE launch (round 4: 42 pack launches of ~9 us each were 0.37 ms of a 21 ms step).  The jobs travel by
// value in the kernel arguments; blockIdx.y = job, grid-stride over its elements.  mode 0 / 1: the row-major packs of
// conv_bf16s.hip (forward / data-grad), mode 2: the tile-major pack above.
struct PackJobDev { const float* w; __bf16* out; unsigned long long total; int Cout, Cin, Kpad, ncs, mode, dgrad, mf16, bn, col; };
struct PackJobsDev { PackJobDev j[CVK_PACK_BATCH_MAX]; };

__global__ void k_pack_batch(const PackJobsDev jobs) {
    const PackJobDev& J = jobs.j[blockIdx.y];
    const float* __restrict__ w = J.w;
    __bf16* __restrict__ out = J.out;
    // one thread = eight consecutive packed elements = one 16-byte store (every pack's length is a multiple of 8: Kpad % 32 == 0).
    // Element by element (one 2-byte store and a div/mod chain per element) the 46 packs of a UNet step took 197 us.
    const size_t total8 = J.total >> 3;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < total8; c += (size_t)gridDim.x * blockDim.x) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (J.mode == 2) {
            const int p = (int)(c & 3), n = (int)((c >> 2) & (J.bn - 1));
            size_t rest = (c >> 2) / J.bn;
            const int tpos = (int)(rest % 9); rest /= 9;
            const int tap = J.col ? (tpos % 3) * 3 + tpos / 3 : tpos;
            const int cs = (int)(rest % J.ncs);
            const int row = (int)(rest / J.ncs) * J.bn + n;
            const int k0 = cs * CK + ((p ^ (J.mf16 ? ((n >> 2) & 1) << 1 : (n >> 2) & 3)) << 3);
            if (!J.dgrad) {
                if (row < J.Cout) {
                    const float* src = w + ((size_t)row * 9 + tap) * J.Cin + k0;
                    if (k0 + 8 <= J.Cin && (J.Cin & 3) == 0) {
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (k0 + e < J.Cin) v[e] = src[e];
                    }
                }
            } else if (row < J.Cin) {
                const float* src = w + ((size_t)k0 * 9 + (8 - tap)) * J.Cin + row;
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k0 + e < J.Cout) v[e] = src[(size_t)e * 9 * J.Cin];
            }
        } else {
            const size_t i0 = c << 3;
            const int k0 = (int)(i0 % J.Kpad);                   // Kpad % 8 == 0: the eight elements share row and tap
            const size_t rt = i0 / J.Kpad;
            const int tap = (int)(rt % 9), row = (int)(rt / 9);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                if (J.mode == 0) { if (row < J.Cout && k < J.Cin) v[e] = w[((size_t)row * 9 + tap) * J.Cin + k]; }
                else             { if (k < J.Cout && row < J.Cin) v[e] = w[((size_t)k * 9 + (8 - tap)) * J.Cin + row]; }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8*>(out + (c << 3)) = o;
    }
}

}  // namespace

namespace cvk_bf16p {

int kind(int Cin, int Cout) {
    const int on = cvk_knob("CVK_BF16P", 1);             // experiments build: 0 = the round-2/3 tile kernels of conv_bf16s.hip for every layer
    const int minci = cvk_knob("CVK_BF16P_MINCI", 64);
    const int h64 = cvk_knob("CVK_BF16P_H64", 1);
    if (!on || Cin < minci || Cin % CK != 0) return 0;
    if (Cout > 64) return 1;
    return (h64 && Cout > 32) ? 2 : 0;
}

bool serves(int Cin, int Cout) { return kind(Cin, Cout) != 0; }

// Kernel-column phases (k_conv_bf16c, round 5) and the column-major weight stream they read: EXPERIMENTS BUILD ONLY (CVK_BF16P_COL=1).
// Built to test VERDICT r4 #1b's hypothesis that k_conv_bf16q loses its time at the barrier pair per 512 matrix cycles: with a third of
// the barriers (and 24 instead of 36 fragment reads per 96 MFMAs) the 23-layer set of configs[3] runs within 0.4 % of k_conv_bf16q
// (2333 against 2342 us over its launches, single layers -3 ... +1 %; parity green on tests/test_gpu_bf16.py) — the barrier cadence is
// NOT what bounds these kernels.  The product keeps the simpler kernel.
#ifdef CVK_EXPERIMENTS
static bool col_major() { return cvk_knob("CVK_BF16P_COL", 0) != 0; }
#else
static constexpr bool col_major() { return false; }
#endif

int stat_partials(int N, int H, int W) { return N * cvk_cdiv(H, TH) * cvk_cdiv(W, TW); }

void pack(const float* w, void* out, int Cout, int Cin, int Kpad, bool dgrad, hipStream_t s) {
    const int rows = dgrad ? Cin : Cout;
    const int k = kind(Kpad, rows);
    const int bn = k == 2 ? HBN : BN;
    const int ntile = cvk_cdiv(rows, bn), ncs = Kpad / CK;
    const size_t total = (size_t)ntile * ncs * 9 * bn * CK;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_pack_w_pp, dim3(blocks), dim3(256), 0, s, w, (__bf16*)out, Cout, Cin, ntile, ncs, dgrad ? 1 : 0, 1, bn, (k == 1 && col_major()) ? 1 : 0);
}

void pack_batch(const cvk_pack_job* jobs, int n, hipStream_t s) {
    PackJobsDev d;
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        const cvk_pack_job& q = jobs[i];
        PackJobDev& o = d.j[i];
        const int rows = q.dgrad ? q.Cin : q.Cout;             // rows of the packed filter; K = the other channel count, padded to Kpad
        const int k = kind(q.Kpad, rows);
        o.w = q.w; o.out = (__bf16*)q.out; o.Cout = q.Cout; o.Cin = q.Cin; o.Kpad = q.Kpad; o.dgrad = q.dgrad ? 1 : 0;
        o.ncs = q.Kpad / CK;
        if (k != 0) {
            o.mode = 2; o.bn = k == 2 ? HBN : BN; o.mf16 = 1; o.col = (k == 1 && col_major()) ? 1 : 0;
            o.total = (unsigned long long)cvk_cdiv(rows, o.bn) * o.ncs * 9 * o.bn * CK;
        } else {
            o.mode = q.dgrad ? 1 : 0; o.bn = 0; o.mf16 = 0; o.col = 0;
            o.total = (unsigned long long)cvk_bf16s_rows_pad(rows) * 9 * q.Kpad;
        }
        if (o.total > most) most = o.total;
    }
    const size_t most8 = most / 8;
    const int bx = (int)((most8 + 255) / 256 < 2048 ? (most8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_pack_batch, dim3(bx > 0 ? bx : 1, n), dim3(256), 0, s, d);
}

static int device_cus() {
    static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 0 ? n : 256; }();
    return cus;
}

int choose(int N, int H, int W, int Cin, int Cout) {
    const int knd = kind(Cin, Cout);
    if (knd != 1) return knd;
    // 128-channel tiles that fill the chip badly (e.g. 384 tiles = 1.5 rounds of 256 CUs): the 64-channel kernel on the same pack
    // (twice the tiles; its K loop and its doubled weight traffic cost a few per cent: it must win 10 % in balance; measured: 1380 tiles -> 2760 is a wash, 384 -> 768 gains 8-9 %)
    const int h128 = cvk_knob("CVK_BF16P_H128", 1);
    const int wg = device_cus();          // NOT the data-parallel cap: the choice (and with it the statistics' summation order) must not depend on it
    const int tilesX = cvk_cdiv(W, TW), tilesY = cvk_cdiv(H, TH);
    const long t1 = (long)N * tilesX * tilesY * cvk_cdiv(Cout, BN), t2 = (long)N * tilesX * tilesY * cvk_cdiv(Cout, HBN);
    const double e1 = (double)t1 / ((double)cvk_cdiv(t1, wg) * wg), e2 = (double)t2 / ((double)cvk_cdiv(t2, wg) * wg);
    return (h128 == 2 || (h128 == 1 && Cout % HBN == 0 && e2 * 0.90 > e1)) ? 3 : 1;
}

void launch(const void* x, const void* wpp, const float* bias, void* y, float* stats, float* counts, int N, int H, int W, int Cin,
            int Cout, int ldy, hipStream_t s, int max_workgroups) {
    // streaming hint on the result stores (csrc/elem_bf16.hip: the result is read once by the pass that follows and must not push the
    // input tiles and weights the neighbouring tiles still need out of L2): +0.9 % on configs[3] (190.9 -> 192.6 img/s, interleaved runs)
    const int nts = cvk_knob("CVK_STREAM_HINTS", 5) >= 1 ? 1 : 0;
    const int tilesX = cvk_cdiv(W, TW), tilesY = cvk_cdiv(H, TH);
    const int ch = choose(N, H, W, Cin, Cout);
    const int knd = ch == 3 ? 2 : ch;
    const bool p128 = ch == 3;
    const int cus = device_cus();
    const int tilesN = cvk_cdiv(Cout, knd == 2 ? HBN : BN);
    const int P = N * tilesX * tilesY;
    const int ntiles = P * tilesN;
    dim3 block(512);
    const int wgcap = max_workgroups > 0 && max_workgroups < cus ? max_workgroups : cus;      // data parallel: CUs left to RCCL
    int g = wgcap;
    {
        const int cap = cvk_knob("CVK_BF16P_GRID", 0);      // experiments build: fewer workgroups (timing)
        if (cap > 0) g = cap;
    }
    if (g > ntiles) g = ntiles;
    dim3 pgrid((unsigned)g);
#define CVK_PP_ARGS (const __bf16*)x, (const char*)wpp, bias, (__bf16*)y, stats, counts, H, W, Cin, Cout, ldy, tilesX, tilesY, tilesN, P, ntiles, nts
    if (knd == 2) {
#ifdef CVK_EXPERIMENTS
        if (p128 && col_major()) {
            if (stats) hipLaunchKernelGGL((k_conv_bf16h<true, 0, true, true>), pgrid, block, 0, s, CVK_PP_ARGS);
            else hipLaunchKernelGGL((k_conv_bf16h<false, 0, true, true>), pgrid, block, 0, s, CVK_PP_ARGS);
            return;
        }
#endif
#ifdef CVK_EXPERIMENTS
        if (p128 && cvk_knob("CVK_BF16H_PIPE", 0) && Cin % 64 == 0 && Cout <= 1024) {
            if (stats) hipLaunchKernelGGL((k_conv_bf16hp<true, true>), pgrid, block, 0, s, CVK_PP_ARGS);
            else hipLaunchKernelGGL((k_conv_bf16hp<false, true>), pgrid, block, 0, s, CVK_PP_ARGS);
            return;
        }
#endif
        if (p128) {
            if (stats) hipLaunchKernelGGL((k_conv_bf16h<true, 0, true>), pgrid, block, 0, s, CVK_PP_ARGS);
            else hipLaunchKernelGGL((k_conv_bf16h<false, 0, true>), pgrid, block, 0, s, CVK_PP_ARGS);
            return;
        }
#ifdef CVK_EXPERIMENTS
        const int hdbg = cvk_knob("CVK_BF16H_DBG", 0);      // per-tile time stamps (tools/tile_stamps_h.py)
        if (hdbg == 2 && stats) { hipLaunchKernelGGL((k_conv_bf16h<true, 1>), pgrid, block, 0, s, CVK_PP_ARGS); return; }
        if (hdbg == 1) { hipLaunchKernelGGL((k_conv_bf16h<false, 1>), pgrid, block, 0, s, CVK_PP_ARGS); return; }
#endif
#ifdef CVK_EXPERIMENTS
        if (stats && !p128 && cvk_knob("CVK_BF16H_MST", 1) == 0) {      // A/B: the round-4 vector-unit statistics
            hipLaunchKernelGGL((k_conv_bf16h<true, 0, false, false, false>), pgrid, block, 0, s, CVK_PP_ARGS);
            return;
        }
#endif
#ifdef CVK_EXPERIMENTS
        if (cvk_knob("CVK_BF16H_PIPE", 0) && Cin % 64 == 0 && Cout <= 1024) {        // pipelined epilogue (round 5 experiment)
            if (stats) hipLaunchKernelGGL((k_conv_bf16hp<true, false>), pgrid, block, 0, s, CVK_PP_ARGS);
            else hipLaunchKernelGGL((k_conv_bf16hp<false, false>), pgrid, block, 0, s, CVK_PP_ARGS);
            return;
        }
#endif
        if (stats) hipLaunchKernelGGL((k_conv_bf16h<true>), pgrid, block, 0, s, CVK_PP_ARGS);
        else hipLaunchKernelGGL((k_conv_bf16h<false>), pgrid, block, 0, s, CVK_PP_ARGS);
        return;
    }
#ifdef CVK_EXPERIMENTS
    if (col_major()) {
        if (stats) hipLaunchKernelGGL((k_conv_bf16c<true>), pgrid, block, 0, s, CVK_PP_ARGS);
        else hipLaunchKernelGGL((k_conv_bf16c<false>), pgrid, block, 0, s, CVK_PP_ARGS);
        return;
    }
#endif
    if (stats) hipLaunchKernelGGL((k_conv_bf16q<true>), pgrid, block, 0, s, CVK_PP_ARGS);
    else hipLaunchKernelGGL((k_conv_bf16q<false>), pgrid, block, 0, s, CVK_PP_ARGS);
#undef CVK_PP_ARGS
}

}  // namespace cvk_bf16p
