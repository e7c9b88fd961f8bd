// Microbenchmark: what does a v_mfma_f32_32x32x2_f32 loop (64 MFMAs per "K step" per wave, 4 accumulators, 256 threads,
// 2 blocks/CU — the shape of the conv GEMM inner loops) reach on this part when the work of a staging pipeline is added
// piece by piece?   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/micro/mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LDT = 36;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)

// MODE bits: 1 = operands re-read from LDS (ds_read_b128 per 16 MFMAs), 2 = barrier per step, 4 = NV VALU fma per step,
//            8 = NL 16-byte buffer loads per step (L2-resident), 16 = 8 ds_write_b128 per step,
//            256 = NL LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB each) per step, all at the top (512: one per 8 MFMAs)
template <int MODE, int NV, int NL>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 256 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * 256 * LDT; i += 256) smem[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float* ar = smem + ((wave >> 1) * 64 + li) * LDT + lh * 4;
    const float* br = smem + 128 * LDT + ((wave & 1) * 64 + li) * LDT + lh * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 24, 0x00020000);
    f32x4 a[2], b[2];
    for (int t = 0; t < 2; ++t) { a[t] = *(const f32x4*)(ar + t * 32 * LDT); b[t] = *(const f32x4*)(br + t * 32 * LDT); }
    f32x4 ld[NL > 0 ? NL : 1];
    for (int i = 0; i < (NL > 0 ? NL : 1); ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    f32x2 pv[8];
    for (int q = 0; q < 8; ++q) pv[q] = f32x2{(float)q, (float)q + 0.5f};
    const f32x2 pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
    unsigned off = (unsigned)(blockIdx.x * 256 + tid) * 16u;
    const unsigned lds_dma_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + 256 * LDT) + wave * 2048);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if ((MODE & 256) && kk == 0) {
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    const char* g = (const char*)src + ((off + q * 65536u) & 0xFFFFF0u);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_dma_base + (q & 1) * 1024) : "memory");
                }
                off += 4096u;
            }
            if ((MODE & 512)) {
#pragma unroll
                for (int q = kk * (NL / 4); q < (kk + 1) * (NL / 4); ++q) {
                    const char* g = (const char*)src + ((off + q * 65536u) & 0xFFFFF0u);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_dma_base + (q & 1) * 1024) : "memory");
                }
                if (kk == 3) off += 4096u;
            }
            if (MODE & 1) {
#pragma unroll
                for (int t = 0; t < 2; ++t) { a[t] = *(const f32x4*)(ar + t * 32 * LDT + kk * 8); b[t] = *(const f32x4*)(br + t * 32 * LDT + kk * 8); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
            if (MODE & 4) {
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);   /* 8 independent chains */
            }
            if (MODE & 128) {
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) pv[q & 7] = __builtin_elementwise_fma(pv[q & 7], pc, pd);   /* v_pk_fma_f32 */
            }
            if ((MODE & 8) && kk == 1) {
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    ld[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (off + q * 65536u) & 0xFFFFF0u, 0, 0));
                }
                off += 4096u;
            }
            if ((MODE & 32) && kk == 0) {
#pragma unroll
                for (int q = 0; q < NL; ++q) v[q & 7] += ld[q][0];
            }
            if ((MODE & 64) && kk == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x4 w = {v[q], v[q], v[q], v[q]};
                    *(f32x4*)(smem + 256 * LDT + ((tid >> 3) + q * 32) * LDT + (tid & 7) * 4) = w;
                }
            }
            if ((MODE & 16) && kk == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x4 w = ld[q % (NL > 0 ? NL : 1)];
                    w[0] += v[q];
                    *(f32x4*)(smem + 256 * LDT + ((tid >> 3) + q * 32) * LDT + (tid & 7) * 4) = w;
                }
            }
        }
        if (MODE & (256 | 512)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 2; ++b_) for (int r = 0; r < 16; ++r) s += acc[a_][b_][r];
    for (int q = 0; q < 8; ++q) s += v[q] + pv[q][0] + pv[q][1];
    for (int q = 0; q < (NL > 0 ? NL : 1); ++q) s += ld[q][0];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NV, int NL>
void run(const char* name, float* out, const float* src) {
    const int iters = 2000, blocks = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, NV, NL>), dim3(blocks), dim3(256), 0, 0, out, src, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, NV, NL>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double flops = (double)blocks * 4 * iters * 64 * (32.0 * 32 * 2 * 2);
    printf("%-58s %.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
int main() {
    float *out, *src;
    if (hipMalloc(&out, 512 * 256 * 4) != hipSuccess || hipMalloc(&src, 1 << 24) != hipSuccess) return 1;
    if (hipMemset(src, 0, 1 << 24) != hipSuccess) return 1;
    run<0, 0, 0>("registers only", out, src);
    run<1, 0, 0>("+ LDS operand reads", out, src);
    run<3, 0, 0>("+ LDS reads + barrier per step", out, src);
    run<7, 64, 0>("+ 64 VALU fma per step", out, src);
    run<7, 256, 0>("+ 256 VALU fma per step", out, src);
    run<7, 512, 0>("+ 512 VALU fma per step", out, src);
    run<3 + 128, 256, 0>("+ 256 v_pk_fma_f32 per step", out, src);
    run<3 + 128, 128, 0>("+ 128 v_pk_fma_f32 per step", out, src);
    run<11, 0, 12>("+ 12 buffer loads per step", out, src);
    run<11, 0, 20>("+ 20 buffer loads per step", out, src);
    run<3 + 8 + 32, 0, 20>("+ 20 loads consumed next step by 20 VALU", out, src);
    run<3 + 8 + 32, 0, 8>("+ 8 loads consumed next step by 8 VALU", out, src);
    run<3 + 64, 0, 0>("+ 8 ds_write_b128 of registers (no loads)", out, src);
    run<27, 0, 20>("+ 20 loads + 8 ds_write_b128 per step", out, src);
    run<31, 128, 20>("+ 20 loads + 8 ds_write + 128 VALU per step", out, src);
    run<31, 256, 20>("+ 20 loads + 8 ds_write + 256 VALU per step", out, src);
    run<27, 0, 8>("+ 8 loads + 8 ds_write_b128 per step", out, src);
    run<3 + 256, 0, 8>("+ 8 LDS-DMA pieces per step (all at the top)", out, src);
    run<3 + 512, 0, 8>("+ 8 LDS-DMA pieces per step (2 per 16 MFMAs)", out, src);
    run<3 + 256, 0, 4>("+ 4 LDS-DMA pieces per step (all at the top)", out, src);
    return 0;
}
