// Microbenchmark: what does a v_mfma_f32_32x32x16_bf16 stream reach on this part next to the LDS operand reads of the
// bf16 convolution kernels?  Per wave and k-step: NM MFMAs on NM independent accumulators, NR LDS reads
// (MODE 1: ds_read_b64_tr_b16 pairs, MODE 2: ds_read_b128), order pinned as in csrc/conv_bf16s.hip (MFMA, then the
// read of the same operand for the next k-step).  256 workgroups x WAVES waves, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16_lds tools/micro/mfma_bf16_lds.hip && /tmp/mfma_bf16_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bf16x8 tr_read8(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 512));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int WAVES, int NM, int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16 * 1024; i += WAVES * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3C003C00u + (i & 255);
    __syncthreads();
    // conflict-free lane address pattern of the weight-grad kernel (128-byte rows, chunk ^ 4*((row>>1)&1))
    const int h = lane >> 5, tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
    const int rl = 8 * h + tq, col = 16 * g1 + 4 * tp;
    const int base_tr = rl * 128 + (((col >> 3) ^ (((rl >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
    const int r = lane & 31;
    const int base_b128 = r * 64 + ((h ^ ((r >> 2) & 3)) << 4);
    f32x16 acc[NM];
    for (int j = 0; j < NM; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 fa, fb[NM];
    auto rd = [&](int slot) -> bf16x8 {
        if (MODE == 1) return tr_read8(smem + base_tr + (slot & 15) * 2048);
        if (MODE == 2) return *reinterpret_cast<const bf16x8*>(smem + base_b128 + (slot & 15) * 2048);
        return bf16x8{};
    };
    fa = MODE ? rd(0) : bf16x8{(__bf16)1.f};
    for (int j = 0; j < NM; ++j) fb[j] = MODE ? rd(j + 1) : bf16x8{(__bf16)(1.f + j)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bf16x8 fan = fa;
            if (MODE) fan = rd(ks + it);
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE) fb[j] = rd(ks + j + it);
                __builtin_amdgcn_sched_barrier(0);
            }
            fa = fan;
        }
    }
    float s = 0.f;
    for (int j = 0; j < NM; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int WAVES, int NM, int MODE> int run(const char* name) {
    float* out; CK(hipMalloc(&out, 256 * WAVES * 64 * 4));
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<WAVES, NM, MODE>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * WAVES * iters * 8 * NM * 2.0 * 32 * 32 * 16;
    printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    CK(hipFree(out));
    return 0;
}

int main() {
    run<4, 9, 0>("4 waves, 9 MFMA/k-step, no LDS");
    run<4, 9, 1>("4 waves, 9 MFMA + 20 tr reads");
    run<4, 9, 2>("4 waves, 9 MFMA + 10 b128 reads");
    run<8, 5, 0>("8 waves, 5 MFMA/k-step, no LDS");
    run<8, 5, 1>("8 waves, 5 MFMA + 12 tr reads");
    run<8, 5, 2>("8 waves, 5 MFMA + 6 b128 reads");
    run<8, 8, 0>("8 waves, 8 MFMA/k-step, no LDS");
    run<8, 8, 2>("8 waves, 8 MFMA + 9 b128 reads");
    return 0;
}
