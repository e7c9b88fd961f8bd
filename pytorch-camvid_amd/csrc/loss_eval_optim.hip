// loss_eval_optim.hip — per-pixel softmax cross-entropy (nn.CrossEntropyLoss(): reference train.py:105,130-131),
// the evaluation reductions (argmax train.py:191; intersection/union histograms utils.py:162-190) and a flat fused
// AdamW step (torch.optim.AdamW: train.py:100,133).  NHWC logits rows are [M][ld] with C valid classes; one thread
// per pixel reads its C contiguous logits (a wave covers 64*ld contiguous floats: fully used cache lines).
#include "cvk_common.h"

namespace {

constexpr int CE_ROWS_PER_BLOCK = 1024;  // 4 chunks of 256 pixels per workgroup
constexpr int CE_CHUNK = 256;            // one pixel per thread per chunk
constexpr int CE_MAX_LD = 63;            // widest pixel row staged through LDS ((ld + 1) * 1 KiB of dynamic LDS <= 64 KiB)

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Global <-> LDS movement of one chunk: the chunk's CE_CHUNK pixel rows are `n` contiguous floats in HBM (rows of ld
// floats); every wave instruction moves 64 consecutive float4 (1 KiB, fully used lines).  In LDS a pixel row has pitch
// ld + 1 floats, so the per-thread row walk below (thread = pixel, column c) is bank-conflict free.
__device__ __forceinline__ void ce_chunk_load(const float* __restrict__ g, float* lds, int n, int ld) {
    const int pitch = ld + 1;
    if ((ld & 3) == 0 && ((uintptr_t)g & 15u) == 0) {
        for (int f = threadIdx.x * 4; f < n; f += CE_CHUNK * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + f);
            const int r = f / ld, c = f - r * ld;                  // ld % 4 == 0: the four floats share a row
            float* q = lds + r * pitch + c;
            q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
        }
    } else {
        for (int f = threadIdx.x; f < n; f += CE_CHUNK) {
            const int r = f / ld;
            lds[r * pitch + (f - r * ld)] = g[f];
        }
    }
}

__device__ __forceinline__ void ce_chunk_store(float* __restrict__ g, const float* lds, int n, int ld) {
    const int pitch = ld + 1;
    if ((ld & 3) == 0 && ((uintptr_t)g & 15u) == 0) {
        for (int f = threadIdx.x * 4; f < n; f += CE_CHUNK * 4) {
            const int r = f / ld, c = f - r * ld;
            const float* q = lds + r * pitch + c;
            const f32x4 v = {q[0], q[1], q[2], q[3]};
            *reinterpret_cast<f32x4*>(g + f) = v;
        }
    } else {
        for (int f = threadIdx.x; f < n; f += CE_CHUNK) {
            const int r = f / ld;
            g[f] = lds[r * pitch + (f - r * ld)];
        }
    }
}

// part[b] = sum of -log softmax(logits)[target] over the block's valid pixels, part[nb + b] = number of valid pixels,
// part[2 nb + b] = number of pixels whose target is neither in [0, C) nor ignore_index.
__global__ __launch_bounds__(256) void k_ce_fwd(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                               float* __restrict__ part, int nb, int M, int C, int ignore_index) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const int pitch = ld + 1;
    float acc = 0.f, cnt = 0.f, bad = 0.f;
    const int base = blockIdx.x * CE_ROWS_PER_BLOCK;
    for (int ch = 0; ch < CE_ROWS_PER_BLOCK / CE_CHUNK; ++ch) {
        const int m0 = base + ch * CE_CHUNK;
        if (m0 >= M) break;
        const int rows = min(CE_CHUNK, M - m0);
        __syncthreads();
        ce_chunk_load(logits + (size_t)m0 * ld, lds, rows * ld, ld);
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            const float* p = lds + threadIdx.x * pitch;
            const long t = (long)target[m0 + threadIdx.x];
            if (t != (long)ignore_index) {
                float mx = p[0];
                for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
                float se = 0.f;
                for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
                if (t >= 0 && t < C) {
                    acc += (mx + logf(se)) - p[t];
                    cnt += 1.f;
                } else {
                    bad += 1.f;
                }
            }
        }
    }
    const float s = block_sum_256(acc, red);
    const float n = block_sum_256(cnt, red);
    const float b = block_sum_256(bad, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s;
        part[nb + blockIdx.x] = n;
        part[2 * nb + blockIdx.x] = b;
    }
}

// loss[0] = mean over the valid pixels (NaN when a target was out of range: nn.CrossEntropyLoss raises there),
// loss[1] = number of valid pixels (the backward's divisor), loss[2] = number of out-of-range targets.
__global__ __launch_bounds__(256) void k_ce_finish(const float* __restrict__ part, int nb, float* loss) {
    __shared__ double red[3][256];
    double a = 0.0, n = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        a += (double)part[i];
        n += (double)part[nb + i];
        b += (double)part[2 * nb + i];
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = n; red[2][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = red[2][0] > 0.0 ? __builtin_nanf("") : (float)(red[0][0] / red[1][0]);   // 0/0 = NaN when every pixel is ignored (as torch)
        loss[1] = (float)red[1][0];
        loss[2] = (float)red[2][0];
    }
}

__global__ __launch_bounds__(256) void k_ce_bwd(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                               const float* __restrict__ loss3, const float* __restrict__ grad_out, float scale,
                                               float* __restrict__ dl, int ld_d, int M, int C, int ignore_index) {
    extern __shared__ float lds[];
    const int pitch = ld + 1;
    const float g = (grad_out != nullptr ? *grad_out : 1.f) * scale / loss3[1];
    const int base = blockIdx.x * CE_ROWS_PER_BLOCK;
    for (int ch = 0; ch < CE_ROWS_PER_BLOCK / CE_CHUNK; ++ch) {
        const int m0 = base + ch * CE_CHUNK;
        if (m0 >= M) break;
        const int rows = min(CE_CHUNK, M - m0);
        __syncthreads();
        ce_chunk_load(logits + (size_t)m0 * ld, lds, rows * ld, ld);
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            float* p = lds + threadIdx.x * pitch;
            const long t = (long)target[m0 + threadIdx.x];
            if (t == (long)ignore_index) {
                for (int c = 0; c < ld; ++c) p[c] = 0.f;
            } else {
                float mx = p[0];
                for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
                float se = 0.f;
                for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
                const float inv = 1.f / se;
                for (int c = 0; c < C; ++c) p[c] = (expf(p[c] - mx) * inv - ((long)c == t ? 1.f : 0.f)) * g;
                for (int c = C; c < ld; ++c) p[c] = 0.f;
            }
        }
        __syncthreads();
        if (ld_d == ld) {
            ce_chunk_store(dl + (size_t)m0 * ld_d, lds, rows * ld, ld);
        } else {
            for (int f = threadIdx.x; f < rows * ld_d; f += CE_CHUNK) {
                const int r = f / ld_d, c = f - r * ld_d;
                dl[(size_t)m0 * ld_d + f] = c < ld ? lds[r * pitch + c] : 0.f;
            }
        }
    }
}

// Wide pixel rows (ld > CE_MAX_LD floats, i.e. 64 classes or more): the LDS-staged kernels above would need more than 64 KiB of
// dynamic LDS, so one thread walks its pixel's row in global memory (reference train.py:105 puts no bound on class_num).
__global__ __launch_bounds__(256) void k_ce_fwd_rows(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                    float* __restrict__ part, int nb, int M, int C, int ignore_index) {
    __shared__ float red[4];
    float acc = 0.f, cnt = 0.f, bad = 0.f;
    const int base = blockIdx.x * CE_ROWS_PER_BLOCK;
    for (int m = base + threadIdx.x; m < min(M, base + CE_ROWS_PER_BLOCK); m += CE_CHUNK) {
        const float* p = logits + (size_t)m * ld;
        const long t = (long)target[m];
        if (t == (long)ignore_index) continue;
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        if (t >= 0 && t < C) {
            acc += (mx + logf(se)) - p[t];
            cnt += 1.f;
        } else {
            bad += 1.f;
        }
    }
    const float s = block_sum_256(acc, red);
    const float n = block_sum_256(cnt, red);
    const float b = block_sum_256(bad, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s;
        part[nb + blockIdx.x] = n;
        part[2 * nb + blockIdx.x] = b;
    }
}

__global__ __launch_bounds__(256) void k_ce_bwd_rows(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                    const float* __restrict__ loss3, const float* __restrict__ grad_out, float scale,
                                                    float* __restrict__ dl, int ld_d, int M, int C, int ignore_index) {
    const float g = (grad_out != nullptr ? *grad_out : 1.f) * scale / loss3[1];
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const float* p = logits + (size_t)m * ld;
        float* o = dl + (size_t)m * ld_d;
        const long t = (long)target[m];
        if (t == (long)ignore_index) {
            for (int c = 0; c < ld_d; ++c) o[c] = 0.f;
            continue;
        }
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) o[c] = (expf(p[c] - mx) * inv - ((long)c == t ? 1.f : 0.f)) * g;
        for (int c = C; c < ld_d; ++c) o[c] = 0.f;
    }
}

__global__ void k_argmax(const float* __restrict__ logits, int ld, int64_t* __restrict__ out, int M, int C) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const float* p = logits + (size_t)m * ld;
        float best = p[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = p[c];
            if (v > best || (v != v && best == best)) { best = v; bi = c; }  // first max; NaN wins like ATen
        }
        out[m] = bi;
    }
}

__global__ __launch_bounds__(256) void k_confusion(const int64_t* __restrict__ pred, const int64_t* __restrict__ label,
                                                  unsigned long long* __restrict__ hist, int M, int K, int ignore) {
    extern __shared__ unsigned int h[];  // [3][K]
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const int l = (int)label[m], p = (int)pred[m];
        if (l == ignore) continue;
        if (p >= 0 && p < K) {
            atomicAdd(&h[K + p], 1u);
            if (p == l) atomicAdd(&h[p], 1u);
        }
        if (l >= 0 && l < K) atomicAdd(&h[2 * K + l], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

// uint8 HWC (cv2 BGR order kept) -> float32 NHWC, 4 channels per pixel (3 valid + zero pad): (v/255 - mean[c]) / std[c]
// = reference transforms.ToTensor + Normalize (transforms.py:485-538) with conf/settings.py:8-9 MEAN/STD passed in.
__global__ void k_preprocess_u8(const uint8_t* __restrict__ src, float* __restrict__ dst, long npix, float m0, float m1,
                                float m2, float r0, float r1, float r2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const uint8_t* p = src + i * 3;
        f32x4 v;
        v[0] = ((float)p[0] * (1.f / 255.f) - m0) * r0;
        v[1] = ((float)p[1] * (1.f / 255.f) - m1) * r1;
        v[2] = ((float)p[2] * (1.f / 255.f) - m2) * r2;
        v[3] = 0.f;
        *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    }
}

}  // namespace

extern "C" int cvk_preprocess_u8(const uint8_t* src, float* dst, int N, int H, int W, const float* mean3, const float* std3,
                                 void* stream) {
    CVK_CHECK_ARG(src && dst && mean3 && std3 && N > 0 && H > 0 && W > 0 && cvk_aligned16(dst), "cvk_preprocess_u8: bad arguments");
    CVK_CHECK_ARG(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "cvk_preprocess_u8: zero std");
    const long npix = (long)N * H * W;
    const long b = (npix + 255) / 256;
    hipLaunchKernelGGL(k_preprocess_u8, dim3((int)(b < 8192 ? b : 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, npix, mean3[0],
                       mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
    CVK_LAUNCH_RETURN("cvk_preprocess_u8");
}

extern "C" int cvk_ce_blocks(int M) { return M > 0 ? cvk_cdiv(M, CE_ROWS_PER_BLOCK) : 0; }

extern "C" int cvk_softmax_ce_fwd(const float* logits, int ld, const int64_t* target, float* part, float* loss, int M, int C,
                                  int ignore_index, void* stream) {
    CVK_CHECK_ARG(logits && target && part && loss && M > 0 && C > 0 && ld >= C, "cvk_softmax_ce_fwd: bad arguments");
    const int nb = cvk_ce_blocks(M);
    hipStream_t s = (hipStream_t)stream;
    if (ld <= CE_MAX_LD)
        hipLaunchKernelGGL(k_ce_fwd, dim3(nb), dim3(256), CE_CHUNK * (ld + 1) * sizeof(float), s, logits, ld, target, part, nb, M, C, ignore_index);
    else
        hipLaunchKernelGGL(k_ce_fwd_rows, dim3(nb), dim3(256), 0, s, logits, ld, target, part, nb, M, C, ignore_index);
    hipLaunchKernelGGL(k_ce_finish, dim3(1), dim3(256), 0, s, part, nb, loss);
    CVK_LAUNCH_RETURN("cvk_softmax_ce_fwd");
}

extern "C" int cvk_softmax_ce_bwd(const float* logits, int ld, const int64_t* target, const float* loss3, const float* grad_out,
                                  float scale, float* dlogits, int ld_d, int M, int C, int ignore_index, void* stream) {
    CVK_CHECK_ARG(logits && target && loss3 && dlogits && M > 0 && C > 0 && ld >= C && ld_d >= C, "cvk_softmax_ce_bwd: bad arguments");
    if (ld <= CE_MAX_LD)
        hipLaunchKernelGGL(k_ce_bwd, dim3(cvk_ce_blocks(M)), dim3(256), CE_CHUNK * (ld + 1) * sizeof(float), (hipStream_t)stream, logits, ld,
                           target, loss3, grad_out, scale, dlogits, ld_d, M, C, ignore_index);
    else
        hipLaunchKernelGGL(k_ce_bwd_rows, dim3(cvk_cdiv(M, 256) < 8192 ? cvk_cdiv(M, 256) : 8192), dim3(256), 0, (hipStream_t)stream, logits, ld,
                           target, loss3, grad_out, scale, dlogits, ld_d, M, C, ignore_index);
    CVK_LAUNCH_RETURN("cvk_softmax_ce_bwd");
}

extern "C" int cvk_argmax_channels(const float* logits, int ld, int64_t* out, int M, int C, void* stream) {
    CVK_CHECK_ARG(logits && out && M > 0 && C > 0 && ld >= C, "cvk_argmax_channels: bad arguments");
    const int blocks = cvk_cdiv(M, 256) < 8192 ? cvk_cdiv(M, 256) : 8192;
    hipLaunchKernelGGL(k_argmax, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, out, M, C);
    CVK_LAUNCH_RETURN("cvk_argmax_channels");
}

extern "C" int cvk_confusion_accumulate(const int64_t* pred, const int64_t* label, int64_t* hist, int M, int num_classes,
                                        int ignore_index, void* stream) {
    CVK_CHECK_ARG(pred && label && hist && M > 0 && num_classes > 0 && num_classes <= 4096, "cvk_confusion_accumulate: bad arguments");
    const int blocks = cvk_cdiv(M, 1024) < 1024 ? cvk_cdiv(M, 1024) : 1024;
    hipLaunchKernelGGL(k_confusion, dim3(blocks), dim3(256), 3 * num_classes * sizeof(unsigned int), (hipStream_t)stream, pred, label,
                       (unsigned long long*)hist, M, num_classes, ignore_index);
    CVK_LAUNCH_RETURN("cvk_confusion_accumulate");
}

extern "C" int cvk_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, void* stream) {
    CVK_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "cvk_adamw_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const int64_t b = (n + 255) / 256;
    hipLaunchKernelGGL(k_adamw, dim3((int)(b < 8192 ? b : 8192)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
    CVK_LAUNCH_RETURN("cvk_adamw_step");
}

// ---- library-wide pieces ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void cvk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cvk_version(void) { return CVK_VERSION; }
#ifndef CVK_ABI_HASH
#error "CVK_ABI_HASH is not defined: build through csrc/Makefile (it hashes include/cvk.h)"
#endif
extern "C" uint64_t cvk_abi_hash(void) { return CVK_ABI_HASH; }
extern "C" const char* cvk_last_error_string(void) { return g_err; }
