// loss_eval_optim.hip — per-pixel softmax cross-entropy (nn.CrossEntropyLoss(): reference train.py:105,130-131),
// the evaluation reductions (argmax train.py:191; intersection/union histograms utils.py:162-190) and a flat fused
// AdamW step (torch.optim.AdamW: train.py:100,133).  NHWC logits rows are [M][ld] with C valid classes; one thread
// per pixel reads its C contiguous logits (a wave covers 64*ld contiguous floats: fully used cache lines).
#include "cvk_common.h"

namespace {

constexpr int CE_ROWS_PER_BLOCK = 1024;  // 256 threads x 4 pixels

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_ce_fwd(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                               float* __restrict__ part, int M, int C) {
    __shared__ float red[4];
    float acc = 0.f;
    const int base = blockIdx.x * CE_ROWS_PER_BLOCK;
    for (int r = threadIdx.x; r < CE_ROWS_PER_BLOCK; r += 256) {
        const int m = base + r;
        if (m >= M) break;
        const float* p = logits + (size_t)m * ld;
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        const int t = (int)target[m];
        const float lt = (t >= 0 && t < C) ? p[t] : 0.f;
        acc += (mx + logf(se)) - lt;
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_ce_finish(const float* __restrict__ part, int nb, float* loss, int M) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) a += (double)part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] / (double)M);
}

__global__ __launch_bounds__(256) void k_ce_bwd(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                               const float* __restrict__ grad_out, float scale, float* __restrict__ dl, int ld_d,
                                               int M, int C) {
    const float g = (grad_out != nullptr ? *grad_out : 1.f) * scale / (float)M;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const float* p = logits + (size_t)m * ld;
        float* q = dl + (size_t)m * ld_d;
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        const float inv = 1.f / se;
        const int t = (int)target[m];
        for (int c = 0; c < C; ++c) q[c] = (expf(p[c] - mx) * inv - (c == t ? 1.f : 0.f)) * g;
        for (int c = C; c < ld_d; ++c) q[c] = 0.f;
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
                                  void* stream) {
    CVK_CHECK_ARG(logits && target && part && loss && M > 0 && C > 0 && ld >= C, "cvk_softmax_ce_fwd: bad arguments");
    const int nb = cvk_ce_blocks(M);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ce_fwd, dim3(nb), dim3(256), 0, s, logits, ld, target, part, M, C);
    hipLaunchKernelGGL(k_ce_finish, dim3(1), dim3(256), 0, s, part, nb, loss, M);
    CVK_LAUNCH_RETURN("cvk_softmax_ce_fwd");
}

extern "C" int cvk_softmax_ce_bwd(const float* logits, int ld, const int64_t* target, const float* grad_out, float scale,
                                  float* dlogits, int ld_d, int M, int C, void* stream) {
    CVK_CHECK_ARG(logits && target && dlogits && M > 0 && C > 0 && ld >= C && ld_d >= C, "cvk_softmax_ce_bwd: bad arguments");
    const int blocks = cvk_cdiv(M, 256) < 8192 ? cvk_cdiv(M, 256) : 8192;
    hipLaunchKernelGGL(k_ce_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, target, grad_out, scale, dlogits, ld_d, M, C);
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
extern "C" const char* cvk_last_error_string(void) { return g_err; }
