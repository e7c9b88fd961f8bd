// Microbenchmark: does vector-ALU work issued by ANOTHER wave of the same SIMD run beside the fp32
// MFMAs of a consumer wave, or does it take matrix time exactly like VALU work in the MFMA wave's own stream
// (tools/micro/mfma_peak.hip: +128 VALU per 64 MFMAs -> 73 % of peak)?
//   same<NV, IL>   : 256 threads, 2 workgroups/CU (8 waves/CU); every wave: LDS operand reads + 64 MFMAs + NV v_fma_f32 per
//                    step + barrier.  IL = 0: the FMAs in four bursts, IL = 1: one FMA group after every MFMA (sched_group_barrier)
//   split<NV, ST>  : 512 threads, 1 workgroup/CU (8 waves/CU, one consumer + one producer wave per SIMD); waves 0-3:
//                    LDS operand reads + 128 MFMAs per step, no VALU; waves 4-7: 2*NV v_fma_f32 (+ ST: 20 buffer loads and
//                    8 ds_write_b128) per step; one barrier per step for all eight waves.
// Same MFMA count and same VALU count per CU and step in both shapes.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_xwave tools/micro/mfma_xwave.hip && ./mfma_xwave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDT = 36;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)

template <int NV, int IL>
__global__ __launch_bounds__(256, 2) void k_same(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[256 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 256 * LDT; i += 256) smem[i] = (float)((i * 7 + 3) % 11) * 0.125f - 0.5f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float* ar = smem + ((wave >> 1) * 64 + li) * LDT + lh * 4;
    const float* br = smem + 128 * LDT + ((wave & 1) * 64 + li) * LDT + lh * 4;
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { a[t] = *(const f32x4*)(ar + t * 32 * LDT + kk * 8); b[t] = *(const f32x4*)(br + t * 32 * LDT + kk * 8); }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
            if (IL && NV >= 64) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, NV / 64, 0);
                }
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 2; ++b_) for (int r = 0; r < 16; ++r) s += acc[a_][b_][r];
    for (int q = 0; q < 8; ++q) s += v[q];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NV, int ST>
__global__ __launch_bounds__(512, 2) void k_split(float* out, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 256 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * 256 * LDT; i += 512) smem[i] = (float)((i * 7 + 3) % 11) * 0.125f - 0.5f;
    __syncthreads();
    float s = 0.f;
    if (wave < 4) {                                   // consumers: ds_read + MFMA only
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const float* ar = smem + ((wave >> 1) * 64 + li) * LDT + lh * 4;
        const float* br = smem + 128 * LDT + ((wave & 1) * 64 + li) * LDT + lh * 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                f32x4 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) { a[t] = *(const f32x4*)(ar + t * 32 * LDT + (kk & 3) * 8); b[t] = *(const f32x4*)(br + t * 32 * LDT + (kk & 3) * 8); }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
            }
            __syncthreads();
        }
        for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 2; ++b_) for (int r = 0; r < 16; ++r) s += acc[a_][b_][r];
    } else {                                          // producers: VALU (+ loads + LDS stores), no MFMA
        float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 24, 0x00020000);
        f32x4 ld[20];
        for (int i = 0; i < 20; ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned off = (unsigned)(blockIdx.x * 256 + (tid - 256)) * 16u;
        const int pt = tid - 256;
        for (int it = 0; it < iters; ++it) {
            if (ST) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {         // store last step's loads (transformed), then issue this step's
                    f32x4 w = ld[q] + ld[q + 8];
                    w[0] += v[q];
                    *(f32x4*)(smem + 256 * LDT + ((pt >> 3) + q * 32) * LDT + (pt & 7) * 4) = w;
                }
#pragma unroll
                for (int q = 0; q < 20; ++q)
                    ld[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (off + q * 65536u) & 0xFFFFF0u, 0, 0));
                off += 4096u;
            }
#pragma unroll
            for (int q = 0; q < 2 * NV; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
            __syncthreads();
        }
        for (int q = 0; q < 8; ++q) s += v[q];
        for (int q = 0; q < 20; ++q) s += ld[q][0];
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <typename F>
void timeit(const char* name, int blocks, double mfma_per_block_iter, int iters, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double flops = (double)blocks * iters * mfma_per_block_iter * (32.0 * 32 * 2 * 2);
    printf("%-72s %.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

#define SAME(NV, IL, label) timeit(label, 512, 4 * 64.0, iters, [&] { hipLaunchKernelGGL((k_same<NV, IL>), dim3(512), dim3(256), 0, 0, out, iters); })
#define SPLIT(NV, ST, label) timeit(label, 256, 4 * 128.0, iters, [&] { hipLaunchKernelGGL((k_split<NV, ST>), dim3(256), dim3(512), 0, 0, out, src, iters); })

int main() {
    float *out, *src;
    const int iters = 2000;
    if (hipMalloc(&out, 512 * 512 * 4) != hipSuccess || hipMalloc(&src, 1 << 24) != hipSuccess) return 1;
    if (hipMemset(src, 0, 1 << 24) != hipSuccess) return 1;
    SAME(0, 0, "same wave:  64 MFMA + 0 VALU per wave-step");
    SAME(64, 0, "same wave:  64 MFMA + 64 VALU (4 bursts of 16)");
    SAME(64, 1, "same wave:  64 MFMA + 64 VALU (1 per MFMA)");
    SAME(128, 0, "same wave:  64 MFMA + 128 VALU (4 bursts of 32)");
    SAME(128, 1, "same wave:  64 MFMA + 128 VALU (2 per MFMA)");
    SAME(256, 0, "same wave:  64 MFMA + 256 VALU (4 bursts of 64)");
    SPLIT(0, 0, "split waves: consumer 128 MFMA | producer idle");
    SPLIT(64, 0, "split waves: consumer 128 MFMA | producer 128 VALU");
    SPLIT(128, 0, "split waves: consumer 128 MFMA | producer 256 VALU");
    SPLIT(256, 0, "split waves: consumer 128 MFMA | producer 512 VALU");
    SPLIT(512, 0, "split waves: consumer 128 MFMA | producer 1024 VALU");
    SPLIT(0, 1, "split waves: consumer 128 MFMA | producer 20 loads + 8 ds_write");
    SPLIT(128, 1, "split waves: consumer 128 MFMA | producer 20 loads + 8 ds_write + 256 VALU");
    return 0;
}
