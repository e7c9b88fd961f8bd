// wgradp.hip — transposed Winograd F(4,3) weight-gradient with BOTH transforms outside the GEMM
// (reference: the weight gradient of nn.Conv2d(cin,cout,3,padding=1), models/unet.py:11, backward of train.py:131).
//
//   dW[co][r][0..2][ci] = G^T [ sum_t  E_xi[t][co] * V_xi[t + (r-1) rows][ci] ],   xi = 0..5,  E = A dy,  V = B^T d  (csrc/wino4.hip)
//
// wino4.hip's k_wgrad_wino4 transforms V in its staging path: ~110 vector instructions per 31-62 MFMAs, and on this part a
// wave's vector instructions ADD to its fp32-MFMA time (tools/micro/mfma_xwave.hip): 0.57-0.62 of peak.  Here the two
// operands are written ONCE as transform-domain planes (1.5x the tensor each, HBM-bound passes), in a layout that makes the
// kernel-row shift a constant and every boundary a zero row:
//     plane[xi][ Wtp zero rows | n: (H + 2) image rows x Wtp column groups | Wtp zero rows ][C],  Wtp = ceil(W/4) rounded up to 8
// (image rows 0 and H+1 of every image, and column groups >= ceil(W/4), are zero), and the GEMM has NO vector arithmetic, NO
// validity logic and NO barrier: a workgroup is ONE wave that owns a 64 x 64 (co, ci) block of one transform index for all three
// kernel rows and walks a run of depth steps (one step = the eight column groups of a strip in one image row) DOWN a strip:
// the V block of image row y serves kernel rows 0 / 1 / 2 of the steps y+1 / y / y-1, so a step fetches ONE new V block and
// one E block (4 LDS-DMA pieces of 1 KiB per 96 MFMAs) and keeps the last three V blocks in a ring.  Operands lie in LDS as in
// memory ([depth row][channel]); v_mfma_f32_16x16x4_f32 with interleaved rows / columns: a lane's ds_read_b128 of an E row
// feeds the four 16-row blocks, its ds_read_b128 of a V row the four 16-column blocks -> 4 reads per 48 MFMAs.  Synchronisation
// is the wave's own counted vmcnt.  Two waves per SIMD (192 accumulator registers each), 12 KiB of LDS per wave.
// Partial sums per run go to slabs [run][xi][Cout][3*Cin], summed in a fixed order and transformed with G^T by k_wgradp_reduce
// (deterministic, no atomics).
#include "conv_tile.h"
#include "lds_dma.h"
#include <utility>

namespace {

template <int... Ks, class F>
__device__ __forceinline__ void p_static_for(std::integer_sequence<int, Ks...>, F&& f) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

__device__ __forceinline__ void p_dma16(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// rows of one plane, and the row of (image n, padded image row yp in [0, H+2), column group xt)
__host__ __device__ inline long p_rows(int N, int H, int Wtp) { return (long)N * (H + 2) * Wtp + 2L * Wtp; }

// ---- transform-domain planes -------------------------------------------------------------------------------------------
// MODE 0: V = B^T d of x (six taps d_j = x[.][4 xt - 1 + j]);  MODE 1: E = A dy (four columns dy[.][4 xt + i]).
// One thread = one plane row x 4 channels; pad rows / pad column groups are written as zeros (the planes live in a reused
// workspace).  Reads 1x, writes 1.5x the tensor.
// SM (MODE 0 only): the SLICE-MAJOR plane layout  plane[xi][C / 16][rows][16]  — the layout the fused forward kernel writes from its staging
// path (csrc/wino4f.hip, VPL: a wave's store covers 16 consecutive plane rows x 64 bytes = 1 KiB contiguous); element (xi, row, c) at
// ((xi * (C / 16) + c / 16) * rows + row) * 16 + c % 16.  C % 16 == 0.
template <int MODE, bool SM = false>
__global__ __launch_bounds__(256) void k_wgradp_planes(const float* __restrict__ X, int ld, float* __restrict__ P, int N, int H, int W,
                                                      int Wt, int Wtp, int C) {
    const int cvn = C >> 2;
    const long rows = p_rows(N, H, Wtp);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cvn) return;
    const long row = idx / cvn;
    const int c = (int)(idx - row * cvn) * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[6] = {z, z, z, z, z, z};
    const long q = row - Wtp;                              // (n * (H + 2) + yp) * Wtp + xt
    if (q >= 0 && q < (long)N * (H + 2) * Wtp) {
        const int xt = (int)(q % Wtp);
        const long ry = q / Wtp;
        const int yp = (int)(ry % (H + 2)), n = (int)(ry / (H + 2));
        if (yp >= 1 && yp <= H && xt < Wt) {
            const float* p = X + (((size_t)n * H + (yp - 1)) * W + 4 * (size_t)xt) * ld + c;
            if (MODE == 0) {
                f32x4 d[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int col = 4 * xt - 1 + j;
                    d[j] = (col >= 0 && col < W) ? *reinterpret_cast<const f32x4*>(p + (long)(j - 1) * ld) : z;
                }
                const f32x4 a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], e = d[4] - d[2], f = d[3] - d[1];
                v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
                v[1] = a + b;
                v[2] = a - b;
                v[3] = e + 2.f * f;
                v[4] = e - 2.f * f;
                v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
            } else {
                f32x4 d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = (4 * xt + i < W) ? *reinterpret_cast<const f32x4*>(p + (long)i * ld) : z;
                const f32x4 a = d[0] + d[2], b = d[1] + d[3], cc = d[0] + 4.f * d[2], dd = 2.f * d[1] + 8.f * d[3];
                v[0] = d[0];
                v[1] = a + b;
                v[2] = a - b;
                v[3] = cc + dd;
                v[4] = cc - dd;
                v[5] = d[3];
            }
        }
    }
    float* o = SM ? P + ((size_t)(c >> 4) * rows + (size_t)row) * 16 + (c & 15) : P + (size_t)row * C + c;
    const size_t ps = (size_t)rows * C;
#pragma unroll
    for (int x = 0; x < 6; ++x) *reinterpret_cast<f32x4*>(o + x * ps) = v[x];
}

// zero rows / column groups of the six planes that no producer writes: the Wtp rows before and after, image rows 0 and H+1 of
// every image, column groups >= Wt of every image row (for planes written by cvk_bn_bwd_dx_e6)
template <bool SM = false>
__global__ __launch_bounds__(256) void k_wgradp_zero_pads(float* __restrict__ P, int N, int H, int Wt, int Wtp, int C, int nplanes) {
    const int cvn = C >> 2;
    const long rows = p_rows(N, H, Wtp);
    const int padc = Wtp - Wt;
    // enumerate the pad rows: 2 * Wtp end rows, N * 2 * Wtp border rows, N * H * padc column-group pads
    const long n_end = 2L * Wtp, n_border = (long)N * 2 * Wtp, n_col = (long)N * H * padc;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long pr = idx / cvn;
    const int c = (int)(idx - pr * cvn) * 4;
    if (pr >= n_end + n_border + n_col) return;
    long row;
    if (pr < n_end) {
        row = pr < Wtp ? pr : rows - Wtp + (pr - Wtp);
    } else if (pr < n_end + n_border) {
        const long k = pr - n_end;
        const int n = (int)(k / (2 * Wtp)), rem = (int)(k - (long)n * 2 * Wtp);
        const int yp = rem < Wtp ? 0 : H + 1, xt = rem < Wtp ? rem : rem - Wtp;
        row = (long)Wtp + ((long)n * (H + 2) + yp) * Wtp + xt;
    } else {
        const long k = pr - n_end - n_border;
        const long ry = k / padc;                              // n * H + y
        const int xt = Wt + (int)(k - ry * padc);
        const int n = (int)(ry / H), yy = (int)(ry - (long)n * H);
        row = (long)Wtp + ((long)n * (H + 2) + yy + 1) * Wtp + xt;
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const size_t ps = (size_t)rows * C;
    for (int x = 0; x < nplanes; ++x)
        *reinterpret_cast<f32x4*>(P + x * ps + (SM ? ((size_t)(c >> 4) * rows + (size_t)row) * 16 + (c & 15) : (size_t)row * C + c)) = z;
}

// ---- the GEMM: one wave per workgroup ----------------------------------------------------------------------------------
// task = (transform index xi, co block, ci block, run): the depth steps q = strip * H + y of its run, strip = (n, column-group
// octet).  LDS (12 KiB): V ring of four 2 KiB blocks (image rows y-1, y, y+1 in use, y+2 arriving), E double buffer.
// VSM: the V planes are slice-major (k_wgradp_planes<0, true> / the fused forward kernel): a 64-channel block of 8 plane rows is four
// 512-byte runs (one per 16-channel slice); only the per-lane source offset of the two DMA pieces and the row pitch differ — the LDS
// image ([depth row][64 channels]) and everything behind it are the same.
// EDY (round 6): E6 holds only the four planes E1..E4; E0 and E5 = columns 4 xt and 4 xt + 3 of dy are read from `dy` [N*H*W (+ slack)][Cout] itself —
// the block of (strip, image row y) is eight pixels 4 * Cout floats apart.  Column groups >= ceil(W/4) of a row then read the next row's (finite)
// pixels, or the caller's zeroed slack behind the last row: their V rows are zero.
template <bool VSM, bool EDY>
__global__ __launch_bounds__(64, 2) void k_wgradp_gemm(const float* __restrict__ E6, const float* __restrict__ V6, float* __restrict__ slab,
                                                      int H, int Wtp, int Cin_ld, int Cout, long rows, int Q, int runs, int nci, int nco,
                                                      const float* __restrict__ dy, int W) {
    __shared__ __attribute__((aligned(1024))) char smem[12 * 1024];
    const unsigned smem_addr = cvk_lds_addr(smem);
    const int lane = threadIdx.x;
    const int lj = lane & 15, kq = lane >> 4;

    // tasks of one run are neighbours (same pixels: one L2), transform index fastest
    const int per = 6 * nci * nco;
    const int id = cvk_xcd_remap(blockIdx.x, gridDim.x);
    const int run = id / per, u = id - run * per;
    const int xi = u % 6, cit = (u / 6) % nci, cot = u / (6 * nci);
    const int qb = (int)((long)run * Q / runs), qe = (int)((long)(run + 1) * Q / runs);
    if (qb >= qe) return;
    const int so = Wtp >> 3;                               // strips per image

    const bool edy = EDY && (xi == 0 || xi == 5);
    const float* const Eb = edy ? dy + cot * 64 + (xi == 5 ? 3 * Cout : 0) : E6 + (size_t)(EDY ? xi - 1 : xi) * rows * Cout + cot * 64;
    const int epitch = edy ? 4 * Cout : Cout;              // floats between consecutive E rows
    const float* const Vb = V6 + (size_t)xi * rows * Cin_ld + (VSM ? (size_t)cit * 4 * rows * 16 : (size_t)cit * 64);
    const int vpitch = VSM ? 16 : Cin_ld;                  // floats between consecutive plane rows of V
    // a 2 KiB block = 8 plane rows x 256 B (64 channels): two DMA pieces of 4 rows; lane -> row lane / 16, 16-byte chunk lane % 16
    const unsigned evoff = (unsigned)(((lane >> 4) * epitch + (lane & 15) * 4) * 4);
    const unsigned vvoff = VSM ? (unsigned)((((unsigned long)((lane & 15) >> 2) * (unsigned long)rows + (lane >> 4)) * 16 + (lane & 3) * 4) * 4)
                               : (unsigned)(((lane >> 4) * Cin_ld + (lane & 15) * 4) * 4);
    auto block_row = [&](int strip, int yp) -> long {       // first plane row of (strip, padded image row yp)
        const int n = strip / so, xg = strip - n * so;
        return (long)Wtp + ((long)n * (H + 2) + yp) * Wtp + 8 * xg;
    };
    auto dma_V = [&](int strip, int yp) {                   // -> ring slot yp & 3
        const float* src = Vb + (size_t)block_row(strip, yp) * vpitch;
        const unsigned dst = smem_addr + (yp & 3) * 2048;
        p_dma16(src, vvoff, dst);
        p_dma16(src + 4 * (size_t)vpitch, vvoff, dst + 1024);
    };
    auto dma_E = [&](int strip, int yp, int slot) {
        const int en = strip / so, exg = strip - en * so;
        const float* src = edy ? Eb + (((size_t)en * H + (yp - 1)) * W + 32 * (size_t)exg) * Cout : Eb + (size_t)block_row(strip, yp) * Cout;
        const unsigned dst = smem_addr + 8192 + slot * 2048;
        p_dma16(src, evoff, dst);
        p_dma16(src + 4 * (size_t)epitch, evoff, dst + 1024);
    };

    f32x4 acc[3][16];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < 16; ++b) acc[r][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frag = kq * 256 + lj * 16;                    // depth row kq of a 4-row group, 16-byte chunk lj

    int strip = qb / H, y = qb - strip * H;                 // current step: image row y (padded row y + 1)
    // prologue of a strip: V rows y, y+1, y+2 (padded) and E row y+1; then per step one V block + one E block arrive
    dma_V(strip, y);
    dma_V(strip, y + 1);
    dma_V(strip, y + 2);
    dma_E(strip, y + 1, 0);
    int es = 0;
    for (int q = qb; q < qe; ++q) {
        // issue the next step's blocks (its V rows y+1, y+2 are here; y+3 and E row y+2 are new) — or, at the end of a strip,
        // nothing: the next strip starts with its own prologue below
        const bool last_in_strip = (y == H - 1);
        const bool more = q + 1 < qe;
        // the slot being refilled (V row y+3 -> slot (y+3)&3 = (y-1)&3) was last read in the PREVIOUS step: all of this wave's LDS
        // reads of that step have returned (its MFMAs consumed them) -> only drain the counter before overwriting
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more && !last_in_strip) {
            dma_V(strip, y + 3);
            dma_E(strip, y + 2, es ^ 1);
            cvk_wait_vm<4>();                               // everything older than these four pieces has landed
        } else {
            cvk_wait_vm<0>();
        }
        __builtin_amdgcn_s_setprio(1);                     // the step's 96 MFMAs ahead of the SIMD's other wave's DMA issue / counted wait (round 6:
        const char* const eblk = smem + 8192 + es * 2048;
        const char* const v0 = smem + ((y + 0) & 3) * 2048;      // padded rows y, y+1, y+2 = image rows y-1, y, y+1 = kernel rows 0, 1, 2
        const char* const v1 = smem + ((y + 1) & 3) * 2048;
        const char* const v2 = smem + ((y + 2) & 3) * 2048;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(eblk + ks * 1024 + frag);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(v0 + ks * 1024 + frag);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(v1 + ks * 1024 + frag);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(v2 + ks * 1024 + frag);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b0[j], acc[0][i * 4 + j], 0, 0, 0);
                    acc[1][i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b1[j], acc[1][i * 4 + j], 0, 0, 0);
                    acc[2][i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b2[j], acc[2][i * 4 + j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);                     //  3.61 -> 3.55 ms for the eight launches, 0.808 -> 0.822 executed)
        if (more) {
            if (last_in_strip) {                           // next strip: refill the ring (one exposed DMA round trip per strip)
                ++strip;
                y = 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_V(strip, 0);
                dma_V(strip, 1);
                dma_V(strip, 2);
                dma_E(strip, 1, es ^ 1);
            } else {
                ++y;
            }
            es ^= 1;
        }
    }

    // P block -> slab[run][xi][co][r * Cin_ld + ci]: row block i, register e, lane (lj, kq): co = 4 (4 kq + e) + i;
    // column blocks j = 0..3 at lane lj: ci = 4 lj + j  -> one 16-byte store per (r, i, e)
    const int K3 = 3 * Cin_ld;
    float* const out = slab + ((size_t)(run * 6 + xi) * Cout + cot * 64) * K3 + cit * 64 + 4 * lj;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 4 * (4 * kq + e) + i;
                const f32x4 v = {acc[r][i * 4 + 0][e], acc[r][i * 4 + 1][e], acc[r][i * 4 + 2][e], acc[r][i * 4 + 3][e]};
                *reinterpret_cast<f32x4*>(out + (size_t)co * K3 + r * Cin_ld) = v;
            }
}

// dw[co][r][s][ci] from the slabs: P_xi = sum over runs in a FIXED order, then G^T (same arithmetic as wino4.hip's reduce).
// The slabs are large (2048 waves x 48 KiB = 100 MB for a 64 -> 64 layer) and the outputs few (Cout * 3 * Cin): a block owns 64
// consecutive outputs, its 16 thread groups sum the runs s = g, g + 16, ... (coalesced 256-byte rows), LDS combines the 16
// partials in group order — bitwise reproducible, ~5 TB/s instead of the 0.9 TB/s of one thread per output.
__global__ __launch_bounds__(1024) void k_wgradp_reduce(const float* __restrict__ slab, float* __restrict__ dw, int runs, int Cout, int Cin,
                                                       int Cin_pad) {
    __shared__ float red[6][16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t total = (size_t)Cout * 3 * Cin;
    const size_t i = (size_t)blockIdx.x * 64 + o;
    const size_t plane = (size_t)Cout * 3 * Cin_pad;
    const bool ok = i < total;
    const int ci = ok ? (int)(i % Cin) : 0;
    const size_t cr = ok ? i / Cin : 0;  // co*3 + r
    const float* p = slab + cr * Cin_pad + ci;
    float P[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int s = g; s < runs; s += 16) {
#pragma unroll
            for (int x = 0; x < 6; ++x) P[x] += p[((size_t)s * 6 + x) * plane];
        }
#pragma unroll
    for (int x = 0; x < 6; ++x) red[x][g][o] = P[x];
    __syncthreads();
    if (g == 0 && ok) {
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) a += red[x][k][o];
            P[x] = a;
        }
        const float s12 = P[1] + P[2], d12 = P[2] - P[1], s34 = P[3] + P[4], d34 = P[3] - P[4];
        float* out = dw + (cr * 3) * Cin + ci;
        out[0] = 0.25f * P[0] - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
        out[Cin] = d12 * (1.f / 6.f) + d34 * (1.f / 12.f);
        out[2 * (size_t)Cin] = (s34 - s12) * (1.f / 6.f) + P[5];
    }
}

struct PPlan { int Wt, Wtp, nci, nco, Q, runs; long rows; };
PPlan plan_wgradp(int N, int H, int W, int Cin_ld, int Cout) {
    PPlan p;
    p.Wt = (W + 3) / 4;
    p.Wtp = (p.Wt + 7) / 8 * 8;
    p.nci = Cin_ld / 64;
    p.nco = Cout / 64;
    p.rows = p_rows(N, H, p.Wtp);
    p.Q = N * (p.Wtp / 8) * H;                           // depth steps
    // two waves per SIMD on 256 CUs = 2048 resident single-wave workgroups: one round of them (equal runs), fewer when a run
    // would be shorter than 16 steps
    const int per = 6 * p.nci * p.nco;
    int runs = 2048 / per;
    if (runs < 1) runs = 1;
    if (runs > p.Q / 16) runs = p.Q / 16 > 0 ? p.Q / 16 : 1;
    p.runs = runs;
    return p;
}

}  // namespace

extern "C" long cvk_wgradp_plane_rows(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return p_rows(N, H, ((W + 3) / 4 + 7) / 8 * 8);
}

// zero the pad rows of six planes [6][cvk_wgradp_plane_rows][C] that cvk_bn_bwd_dx_e6 is about to fill
extern "C" int cvk_wgradp_zero_pads(float* planes, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(planes && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && cvk_aligned16(planes), "cvk_wgradp_zero_pads: bad arguments");
    const int Wt = (W + 3) / 4, Wtp = (Wt + 7) / 8 * 8;
    const long pr = 2L * Wtp + (long)N * 2 * Wtp + (long)N * H * (Wtp - Wt);
    const long threads = pr * (C / 4);
    hipLaunchKernelGGL(k_wgradp_zero_pads<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, planes, N, H, Wt, Wtp, C, 6);
    CVK_LAUNCH_RETURN("cvk_wgradp_zero_pads");
}

// ... of the FOUR planes E1..E4 that cvk_bn_bwd_dx_e4p is about to fill
extern "C" int cvk_wgradp_zero_pads4(float* planes, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(planes && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && cvk_aligned16(planes), "cvk_wgradp_zero_pads4: bad arguments");
    const int Wt = (W + 3) / 4, Wtp = (Wt + 7) / 8 * 8;
    const long pr = 2L * Wtp + (long)N * 2 * Wtp + (long)N * H * (Wtp - Wt);
    const long threads = pr * (C / 4);
    hipLaunchKernelGGL(k_wgradp_zero_pads<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, planes, N, H, Wt, Wtp, C, 4);
    CVK_LAUNCH_RETURN("cvk_wgradp_zero_pads4");
}

// ... of six SLICE-MAJOR V planes [6][C / 16][rows][16] that the fused forward kernel is about to fill (cvk_conv3x3_wino4f_vplanes)
extern "C" int cvk_wgradp_zero_pads_sm(float* planes, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(planes && N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0 && cvk_aligned16(planes), "cvk_wgradp_zero_pads_sm: bad arguments");
    const int Wt = (W + 3) / 4, Wtp = (Wt + 7) / 8 * 8;
    const long pr = 2L * Wtp + (long)N * 2 * Wtp + (long)N * H * (Wtp - Wt);
    const long threads = pr * (C / 4);
    hipLaunchKernelGGL(k_wgradp_zero_pads<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, planes, N, H, Wt, Wtp, C, 6);
    CVK_LAUNCH_RETURN("cvk_wgradp_zero_pads_sm");
}

extern "C" size_t cvk_conv3x3_wgradp_workspace_bytes(int N, int H, int W, int Cin_ld, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_ld < 64 || Cout < 64 || Cin_ld % 64 || Cout % 64) return 0;
    const PPlan p = plan_wgradp(N, H, W, Cin_ld, Cout);
    return ((size_t)6 * p.rows * (Cin_ld + Cout) + (size_t)p.runs * 6 * Cout * 3 * Cin_ld) * sizeof(float);
}

// ---- the three steps, separately callable (the engine times them apart) -------------------------------------------------------
// planes [6][cvk_wgradp_plane_rows][C] of x (is_dy == 0: V = B^T d) or dy (is_dy != 0: E = A dy), pad rows included
extern "C" int cvk_wgradp_planes(const float* t, int ld, float* planes, int N, int H, int W, int C, int is_dy, void* stream) {
    CVK_CHECK_ARG(t && planes && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0, "cvk_wgradp_planes: bad arguments");
    CVK_CHECK_ARG(cvk_aligned16(t) && cvk_aligned16(planes), "cvk_wgradp_planes: pointers must be 16-byte aligned");
    const int Wt = (W + 3) / 4, Wtp = (Wt + 7) / 8 * 8;
    const long th = p_rows(N, H, Wtp) * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (is_dy) hipLaunchKernelGGL(k_wgradp_planes<1>, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, t, ld, planes, N, H, W, Wt, Wtp, C);
    else hipLaunchKernelGGL(k_wgradp_planes<0>, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, t, ld, planes, N, H, W, Wt, Wtp, C);
    CVK_LAUNCH_RETURN("cvk_wgradp_planes");
}

// V = B^T d of x as six SLICE-MAJOR planes [6][C / 16][rows][16] (pad rows included) — what cvk_conv3x3_wino4f_vplanes leaves behind, as a pass
// of its own (callers whose forward pass ran another kernel; tests)
extern "C" int cvk_wgradp_planes_sm(const float* x, int ld, float* planes, int N, int H, int W, int C, void* stream) {
    CVK_CHECK_ARG(x && planes && N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0 && ld >= C && ld % 4 == 0, "cvk_wgradp_planes_sm: bad arguments");
    CVK_CHECK_ARG(cvk_aligned16(x) && cvk_aligned16(planes), "cvk_wgradp_planes_sm: pointers must be 16-byte aligned");
    const int Wt = (W + 3) / 4, Wtp = (Wt + 7) / 8 * 8;
    const long th = p_rows(N, H, Wtp) * (C / 4);
    hipLaunchKernelGGL((k_wgradp_planes<0, true>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, planes, N, H, W, Wt, Wtp, C);
    CVK_LAUNCH_RETURN("cvk_wgradp_planes_sm");
}

extern "C" size_t cvk_wgradp_gemm_workspace_bytes(int N, int H, int W, int Cin_ld, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_ld < 64 || Cout < 64 || Cin_ld % 64 || Cout % 64) return 0;
    const PPlan p = plan_wgradp(N, H, W, Cin_ld, Cout);
    return (size_t)p.runs * 6 * Cout * 3 * Cin_ld * sizeof(float);
}

// dw from the planes E6 [6][rows][Cout] and V6 [6][rows][Cin_ld]: GEMM into slabs (workspace) + fixed-order reduction with G^T
static int wgradp_gemm_go(bool vsm, const float* E6, const float* V6, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                          void* workspace, size_t workspace_bytes, void* stream, const float* dy = nullptr) {
    CVK_CHECK_ARG(E6 && V6 && dw && workspace, "cvk_wgradp_gemm: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= Cin_ld && Cin_ld % 64 == 0 && Cout % 64 == 0 && Cout >= 64,
                  "cvk_wgradp_gemm: Cin_ld=%d and Cout=%d must be multiples of 64", Cin_ld, Cout);
    CVK_CHECK_ARG(cvk_aligned16(E6) && cvk_aligned16(V6) && cvk_aligned16(workspace), "cvk_wgradp_gemm: pointers must be 16-byte aligned");
    const PPlan p = plan_wgradp(N, H, W, Cin_ld, Cout);
    if (workspace_bytes < cvk_wgradp_gemm_workspace_bytes(N, H, W, Cin_ld, Cout)) {
        cvk_set_error("cvk_wgradp_gemm: workspace too small");
        return CVK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* slab = (float*)workspace;
    const int per = 6 * p.nci * p.nco;
    if (vsm) {
        CVK_CHECK_ARG(3L * p.rows * 64 + 4096 < (1L << 32), "cvk_wgradp_gemm_sm: %ld plane rows exceed the 32-bit lane offset of the slice-major V block", p.rows);
        if (dy) hipLaunchKernelGGL((k_wgradp_gemm<true, true>), dim3(per * p.runs), dim3(64), 0, s, E6, V6, slab, H, p.Wtp, Cin_ld, Cout, p.rows, p.Q, p.runs, p.nci, p.nco, dy, W);
        else hipLaunchKernelGGL((k_wgradp_gemm<true, false>), dim3(per * p.runs), dim3(64), 0, s, E6, V6, slab, H, p.Wtp, Cin_ld, Cout, p.rows, p.Q, p.runs, p.nci, p.nco, dy, W);
    } else {
        hipLaunchKernelGGL((k_wgradp_gemm<false, false>), dim3(per * p.runs), dim3(64), 0, s, E6, V6, slab, H, p.Wtp, Cin_ld, Cout, p.rows, p.Q, p.runs, p.nci, p.nco, dy, W);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cvk_set_error("cvk_wgradp_gemm: launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    const size_t total = (size_t)Cout * 3 * Cin;
    hipLaunchKernelGGL(k_wgradp_reduce, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, s, slab, dw, p.runs, Cout, Cin, Cin_ld);
    CVK_LAUNCH_RETURN("cvk_wgradp_gemm");
}

extern "C" int cvk_wgradp_gemm(const float* E6, const float* V6, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return wgradp_gemm_go(false, E6, V6, dw, N, H, W, Cin, Cin_ld, Cout, workspace, workspace_bytes, stream);
}
// the same with SLICE-MAJOR V planes (cvk_conv3x3_wino4f_vplanes / cvk_wgradp_planes_sm); E6 stays row-major [6][rows][Cout]
extern "C" int cvk_wgradp_gemm_sm(const float* E6, const float* V6sm, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return wgradp_gemm_go(true, E6, V6sm, dw, N, H, W, Cin, Cin_ld, Cout, workspace, workspace_bytes, stream);
}

// ... with E0 / E5 read straight from dy: E4p = the four planes E1..E4 [4][rows][Cout] (cvk_wgradp_zero_pads4 + cvk_bn_bwd_dx_e4p), dy = the dense
// [N*H*W][Cout] tensor those passes wrote, followed by cvk_wgradp_dy_slack(W) * Cout ZERO floats (the last row's pad column groups read past the tensor)
extern "C" int cvk_wgradp_dy_slack(int W) { return W > 0 ? 4 * (((W + 3) / 4 + 7) / 8 * 8) - W + 4 : 0; }
extern "C" int cvk_wgradp_gemm_sm_dy(const float* E4p, const float* dy, const float* V6sm, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(dy && cvk_aligned16(dy), "cvk_wgradp_gemm_sm_dy: dy must be a 16-byte aligned pointer");
    // a ragged last column group (W % 4 != 0) has E5 = 0 where dy's next row begins: only whole groups are columns of dy
    CVK_CHECK_ARG(W % 4 == 0, "cvk_wgradp_gemm_sm_dy: W=%d must be a multiple of 4 (use cvk_wgradp_gemm_sm with six planes)", W);
    return wgradp_gemm_go(true, E4p, V6sm, dw, N, H, W, Cin, Cin_ld, Cout, workspace, workspace_bytes, stream, dy);
}

// one call: E6_pre NULL (the E planes are built from dy in the workspace) or the six planes written by cvk_wgradp_zero_pads +
// cvk_bn_bwd_dx_e6 (dy may then be NULL).
extern "C" int cvk_conv3x3_wgradp(const float* x, const float* dy, const float* E6_pre, float* dw, int N, int H, int W, int Cin, int Cin_ld,
                                  int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream) {
    CVK_CHECK_ARG(x && (dy || E6_pre) && dw && workspace, "cvk_conv3x3_wgradp: null pointer");
    CVK_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= Cin_ld, "cvk_conv3x3_wgradp: bad shape");
    CVK_CHECK_ARG(Cin_ld % 64 == 0 && Cout % 64 == 0 && Cout >= 64 && ld_dy >= Cout && ld_dy % 4 == 0,
                  "cvk_conv3x3_wgradp: Cin_ld=%d and Cout=%d must be multiples of 64", Cin_ld, Cout);
    CVK_CHECK_ARG(cvk_aligned16(workspace), "cvk_conv3x3_wgradp: workspace must be 16-byte aligned");
    const size_t need = cvk_conv3x3_wgradp_workspace_bytes(N, H, W, Cin_ld, Cout);
    if (workspace_bytes < need) {
        cvk_set_error("cvk_conv3x3_wgradp: workspace %zu < %zu bytes", workspace_bytes, need);
        return CVK_EWORKSPACE;
    }
    const PPlan p = plan_wgradp(N, H, W, Cin_ld, Cout);
    float* V6 = (float*)workspace;
    float* E6 = V6 + (size_t)6 * p.rows * Cin_ld;
    float* slab = E6 + (size_t)6 * p.rows * Cout;
    int rc = cvk_wgradp_planes(x, Cin_ld, V6, N, H, W, Cin_ld, 0, stream);
    if (rc == CVK_OK && E6_pre == nullptr) rc = cvk_wgradp_planes(dy, ld_dy, E6, N, H, W, Cout, 1, stream);
    if (rc == CVK_OK)
        rc = cvk_wgradp_gemm(E6_pre != nullptr ? E6_pre : E6, V6, dw, N, H, W, Cin, Cin_ld, Cout, slab,
                             cvk_wgradp_gemm_workspace_bytes(N, H, W, Cin_ld, Cout), stream);
    return rc;
}
