// Microbenchmark (round 4): what does the OTHER wave of a SIMD cost a wave that streams v_mfma_f32_32x32x16_bf16?
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) issue NM independent MFMAs per iteration and nothing else;
// waves 4-7 (their SIMD partners) run MODE until a time limit:
//   0 exit at once                      1 LDS-DMA, every lane out of the raw buffer (zero fill, no memory traffic)
//   2 LDS-DMA from an L2-resident MiB   3 buffer_load_dwordx4 into registers (same addresses as 2)
//   4 ds_read_b128                      5 eight v_add_u32
//   6 s_sleep only
// one operation every PERIOD-ish cycles (s_sleep between them; PERIOD 0 = back to back).  Printed: cycles per MFMA of wave 0
// (s_memtime; 32 = the matrix pipe never waits).  tools/tile_stamps_wgrad.py measured ~57 lost matrix cycles per DMA piece in
// k_wgrad_bf16r; this separates the instruction from everything around it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_partner tools/micro/mfma_partner.hip && /tmp/mfma_partner
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ i32x4v raw_rsrc_2g(const void* base) {
    const uintptr_t b = (uintptr_t)base;
    i32x4v r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xFFFF);
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}

// round 5: PRIO = s_setprio of the partner waves (the MFMA waves stay at 0); OWNV = v_add_u32 per MFMA in the MFMA waves' OWN stream;
// M16 = v_mfma_f32_16x16x32_bf16 (16 cycles) instead of 32x32x16 (32 cycles)
template <int MODE, int SLEEP, int OWN, int PRIO = 0, int OWNV = 0, int M16 = 0>     // OWN: the MFMA waves issue one zero-fill DMA per iteration themselves (partners idle when MODE 0)
__global__ __launch_bounds__(512, 2) void k(float* out, const float* src, unsigned long long* stamps, int iters, unsigned long long limit) {
    __shared__ __attribute__((aligned(1024))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16 * 1024; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3C003C00u;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const i32x4v rsrc = raw_rsrc_2g(src);
    if (wave < 4) {
        constexpr int NM = 8;
        f32x16 acc[NM];
        for (int j = 0; j < NM; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        bf16x8 fa, fb;
        for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(1.f + lane); fb[i] = (__bf16)0.5f; }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if (OWN) {
                const unsigned off = 0x80000000u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(off), "s"(rsrc), "s"(lds0 + 32768 + wave * 1024) : "memory");
            }
            unsigned vv = lane;
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                if (M16) {
                    f32x4 c4 = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c4, 0, 0, 0);
                    acc[j][0] = c4[0]; acc[j][1] = c4[1]; acc[j][2] = c4[2]; acc[j][3] = c4[3];
                } else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < OWNV; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vv) : "v"(lane));
            }
            asm volatile("" :: "v"(vv));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int j = 0; j < NM; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
        out[blockIdx.x * 512 + tid] = s;
        if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (MODE == 0) return;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned v = lane, n = 0;
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    const unsigned real_off = ((blockIdx.x & 63) * 16384 + (wave & 3) * 1024 + lane * 16);
    while (__builtin_amdgcn_s_memtime() - t0 < limit) {
        if (MODE == 1 || MODE == 2) {
            const unsigned off = MODE == 1 ? 0x80000000u : real_off + ((n & 3) << 12);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(off), "s"(rsrc), "s"(lds0 + wave * 1024) : "memory");
        } else if (MODE == 3) {
            f32x4 r;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(real_off + ((n & 3) << 12)), "s"(rsrc) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sink += r;
        } else if (MODE == 4) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(smem + lane * 16 + (n & 7) * 1024);
            sink += r;
        } else if (MODE == 5) {
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(lane));
        }
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        ++n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + tid] = (float)v + sink[0] + sink[1] + sink[2] + sink[3];
    if (lane == 0) stamps[1024 + blockIdx.x * 4 + (wave & 3)] = n;
}


// Both waves of every SIMD stream MFMAs (the weight-grad kernel's situation).  Between groups of NM MFMAs, VAR:
//   0 nothing   1 zero-fill LDS-DMA, loop-invariant offset register   2 same, offset through a fresh v_mov_b32 (as hipcc emits it for a select)
//   3 as 2 behind a wave-uniform scalar branch that alternates taken / not taken   4 one v_mov_b32 alone   5 s_nop only
//   6 as 1 but reading an L2-resident MiB
template <int VAR, int NM>
__global__ __launch_bounds__(512, 2) void k2(float* out, const float* src, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(1024))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16 * 1024; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3C003C00u;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const i32x4v rsrc = raw_rsrc_2g(src);
    f32x16 acc[NM];
    for (int j = 0; j < NM; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(1.f + lane); fb[i] = (__bf16)0.5f; }
    unsigned off = VAR == 6 ? ((blockIdx.x & 63) * 16384 + wave * 1024 + lane * 16) : 0x80000000u;
    unsigned vsrc = off;
    asm volatile("" : "+v"(vsrc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 1 || VAR == 6) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(off), "s"(rsrc), "s"(lds0 + wave * 1024) : "memory");
        } else if (VAR == 2) {
            unsigned o;
            asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(vsrc));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(o), "s"(rsrc), "s"(lds0 + wave * 1024) : "memory");
        } else if (VAR == 3) {
            if (__builtin_amdgcn_readfirstlane(it + wave) & 1) {
                unsigned o;
                asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(vsrc));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(o), "s"(rsrc), "s"(lds0 + wave * 1024) : "memory");
            }
        } else if (VAR == 4) {
            unsigned o;
            asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(vsrc));
            asm volatile("" : : "v"(o));
        } else if (VAR == 5) {
            asm volatile("s_nop 0");
        }
#pragma unroll
        for (int j = 0; j < NM; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int j = 0; j < NM; ++j) for (int i = 0; i < 16; ++i) sum += acc[j][i];
    out[blockIdx.x * 512 + tid] = sum;
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR, int NM>
int run2(const char* what, float* out, const float* src, unsigned long long* stamps) {
    const int iters = 32000 / NM;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k2<VAR, NM>), dim3(256), dim3(512), 0, 0, out, src, stamps, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[2048];
    CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    double c = 0, mx = 0;
    for (int b = 0; b < 256; ++b) { double m = 0; for (int w = 0; w < 8; ++w) m = m > (double)h[b * 8 + w] ? m : (double)h[b * 8 + w]; c += m; }
    c /= 256;
    const double per = c / (2.0 * iters * NM);       // two waves share the pipe: matrix cycles per MFMA
    printf("both stream, %d MFMAs per group, %-58s %6.2f cycles/MFMA  (%5.1f lost per group and wave)\n", NM, what, per, (per - 32.0) * NM);
    (void)mx;
    return 0;
}

template <int MODE, int SLEEP, int OWN, int PRIO = 0, int OWNV = 0, int M16 = 0>
int run(const char* what, float* out, const float* src, unsigned long long* stamps) {
    const int iters = 4000;
    const unsigned long long limit = (unsigned long long)iters * 8 * 34;
    CK(hipMemset(stamps, 0, 2048 * 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, SLEEP, OWN, PRIO, OWNV, M16>), dim3(256), dim3(512), 0, 0, out, src, stamps, iters, limit);
    CK(hipDeviceSynchronize());
    unsigned long long h[2048];
    CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    double c = 0, n = 0;
    for (int i = 0; i < 1024; ++i) c += (double)h[i];
    for (int i = 0; i < 1024; ++i) n += (double)h[1024 + i];
    c /= 1024; n /= 1024;
    const double per = c / (iters * 8.0);
    printf("%-62s %6.2f cycles/MFMA", what, per);
    if (n > 0) printf("   partner ops %7.0f (one per %5.0f cycles): %5.1f matrix cycles lost per op", n, c / n, (per - 32.0) * iters * 8.0 / n);
    printf("\n");
    return 0;
}

int main() {
    float *out, *src;
    unsigned long long* stamps;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&src, 4 << 20));
    CK(hipMemset(src, 0, 4 << 20));
    CK(hipMalloc(&stamps, 2048 * 8));
    run<0, 0, 0>("partner idle", out, src, stamps);
    run<6, 2, 0>("partner: s_sleep only", out, src, stamps);
    run<1, 2, 0>("partner: LDS-DMA, all lanes out of range, s_sleep 2", out, src, stamps);
    run<1, 0, 0>("partner: LDS-DMA, all lanes out of range, back to back", out, src, stamps);
    run<2, 2, 0>("partner: LDS-DMA from L2, s_sleep 2", out, src, stamps);
    run<2, 0, 0>("partner: LDS-DMA from L2, back to back", out, src, stamps);
    run<3, 2, 0>("partner: buffer_load_dwordx4 to registers + wait, s_sleep 2", out, src, stamps);
    run<4, 2, 0>("partner: ds_read_b128, s_sleep 2", out, src, stamps);
    run<4, 0, 0>("partner: ds_read_b128, back to back", out, src, stamps);
    run<5, 2, 0>("partner: 8 v_add_u32, s_sleep 2", out, src, stamps);
    run<5, 0, 0>("partner: 8 v_add_u32, back to back", out, src, stamps);
    run<0, 0, 1>("own stream: one zero-fill LDS-DMA per 8 MFMAs, partner idle", out, src, stamps);
    run<6, 2, 1>("own stream: one zero-fill LDS-DMA per 8 MFMAs, partner sleeps", out, src, stamps);
    printf("# round 5: partner priority / vector work in the MFMA wave's own stream / 16x16x32 shape\n");
    run<5, 0, 0, 3>("partner: 8 v_add_u32 back to back, partner s_setprio 3", out, src, stamps);
    run<4, 0, 0, 3>("partner: ds_read_b128 back to back, partner s_setprio 3", out, src, stamps);
    run<0, 0, 0, 0, 1>("own stream: 1 v_add_u32 per MFMA, partner idle", out, src, stamps);
    run<0, 0, 0, 0, 2>("own stream: 2 v_add_u32 per MFMA, partner idle", out, src, stamps);
    run<0, 0, 0, 0, 4>("own stream: 4 v_add_u32 per MFMA, partner idle", out, src, stamps);
    run<0, 0, 0, 0, 6>("own stream: 6 v_add_u32 per MFMA, partner idle", out, src, stamps);
    run<0, 0, 0, 0, 0, 1>("16x16x32: partner idle", out, src, stamps);
    run<5, 0, 0, 0, 0, 1>("16x16x32: partner 8 v_add_u32 back to back", out, src, stamps);
    run<5, 0, 0, 3, 0, 1>("16x16x32: partner 8 v_add_u32 back to back, partner s_setprio 3", out, src, stamps);
    run<0, 0, 0, 0, 1, 1>("16x16x32: own stream 1 v_add_u32 per MFMA", out, src, stamps);
    run<0, 0, 0, 0, 2, 1>("16x16x32: own stream 2 v_add_u32 per MFMA", out, src, stamps);
    run<0, 0, 0, 0, 3, 1>("16x16x32: own stream 3 v_add_u32 per MFMA", out, src, stamps);
    run2<0, 3>("nothing between", out, src, stamps);
    run2<5, 3>("s_nop", out, src, stamps);
    run2<4, 3>("one v_mov_b32", out, src, stamps);
    run2<1, 3>("zero-fill LDS-DMA, invariant offset", out, src, stamps);
    run2<2, 3>("zero-fill LDS-DMA, offset via v_mov_b32", out, src, stamps);
    run2<3, 3>("the same behind an alternating scalar branch", out, src, stamps);
    run2<6, 3>("LDS-DMA from L2, invariant offset", out, src, stamps);
    run2<0, 9>("nothing between", out, src, stamps);
    run2<4, 9>("one v_mov_b32", out, src, stamps);
    run2<1, 9>("zero-fill LDS-DMA, invariant offset", out, src, stamps);
    run2<2, 9>("zero-fill LDS-DMA, offset via v_mov_b32", out, src, stamps);
    run2<6, 9>("LDS-DMA from L2, invariant offset", out, src, stamps);
    return 0;
}
