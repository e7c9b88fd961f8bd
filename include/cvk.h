/*
 * cvk.h — C ABI of libcvk.so: the MI355X (gfx950) kernels behind the CamVid UNet/SegNet training hot path.
 *
 * The upstream reference (weiaicunzai/pytorch-camvid) has NO native code and NO FFI on this path: its hot path
 * is torch.nn modules calling ATen.  The boundary a maintainer binds is therefore the set of torch operators the
 * reference invokes; each entry point below names the reference call site (file:line under the reference root)
 * whose ATen operator it replaces.  INTEGRATION.md shows the ctypes stub and the nn.Module that sits on top.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (torch tensors); the library allocates nothing
 *     persistent and keeps no pointer past return.  Workspaces are caller-provided.
 *   - Every call only enqueues work on `stream` (a hipStream_t passed as void*) and never synchronises.
 *   - Re-entrant, no mutable globals except a thread-local last-error string.
 *   - Return value: 0 on success, CVK_E* (<0) on argument errors, or a positive hipError_t from the launch.
 *   - Activations are fp32 NHWC ("channels_last"): element (n,y,x,c) of a dense tensor lives at
 *     ((n*H + y)*W + x)*ld + c with pixel stride ld >= C.  A *view* (cvk_view) addresses a channel slice and/or a
 *     spatial window of a larger NHWC buffer: element (n,y,x,c) at  ptr + n*sN + y*sY + x*sX + c.
 *   - Conv weights are [Cout][3][3][Cin] (= torch OIHW with channels_last strides, i.e. KRSC).
 *   - Vectorised paths need 16-byte aligned base pointers and ld % 4 == 0; the conv kernels REQUIRE it
 *     (the host pads channel counts to a multiple of 4 with zero channels).
 */
#ifndef CVK_H
#define CVK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVK_VERSION 100          /* 0.1.0 */
#define CVK_OK 0
#define CVK_EINVAL (-1)          /* bad argument (shape, alignment, null pointer) */
#define CVK_EWORKSPACE (-2)      /* workspace too small */
#define CVK_STAT_ROWS 64         /* rows (pixels) summarised by one BN-statistics partial of the conv epilogue */

typedef struct cvk_view {        /* strided NHWC view, strides in floats */
    float*  ptr;                 /* address of element (0,0,0,0) of the view */
    int64_t sN, sY, sX;          /* image, row, pixel strides */
} cvk_view;

typedef struct cvk_viewh {       /* strided NHWC view of a bf16 (or, where stated, fp32) tensor; strides in ELEMENTS */
    void*   ptr;
    int64_t sN, sY, sX;
} cvk_viewh;

int         cvk_version(void);
/* first 64 bits of the SHA-256 of THIS header as the library was compiled against it (csrc/Makefile passes it as
 * -DCVK_ABI_HASH).  A host binding hashes the header it was written against and refuses a library built from another
 * one: entry points keep their names when argument lists change, and ctypes / cgo / JNI stubs do not check arity. */
uint64_t    cvk_abi_hash(void);
const char* cvk_last_error_string(void);

/* ---- layout at the module boundary: logical NCHW tensors of any strides <-> dense NHWC (ld >= C) --------------
 * replaces nothing in the reference (torch tensors are layout-polymorphic); it is the import/export step of
 * models/unet.py:94 `forward(x)` input and :156 return value. Pad channels [C,ld) of dst are written as 0. */
int cvk_import_nchw(const float* src, int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                    float* dst, int ld, int N, int C, int H, int W, void* stream);
int cvk_export_nchw(const float* src, int ld, float* dst, int64_t dN, int64_t dC, int64_t dH, int64_t dW,
                    int N, int C, int H, int W, void* stream);
/* zero channels [0,C) of every pixel of an NHWC frame [N,H,W] that lies OUTSIDE the window
 * [y0,y0+h) x [x0,x0+w): the F.pad of models/unet.py:120-123 written in place into the concat buffer. */
int cvk_zero_frame(cvk_view buf, int N, int H, int W, int C, int y0, int x0, int h, int w, void* stream);

/* ---- conv 3x3, stride 1, zero pad 1 (nn.Conv2d(cin,cout,3,padding=1): models/unet.py:11, models/segnet.py:8) --
 * forward: y[m][co] = bias[co] + sum_{tap,ci} x[m+tap][ci] * w[co][tap][ci]      (implicit GEMM on fp32 MFMA)
 *   x: dense NHWC, ld = Cin (Cin % 4 == 0).  w: [Cout][9][Cin].  y: dense, pixel stride ldy >= Cout, columns
 *   [Cout,ldy) are written as 0.  bias may be NULL.
 *   stats (nullable): fused BatchNorm statistics partials, float[2][P][Cout], P = ceil(N*H*W / CVK_STAT_ROWS):
 *   stats[0][p][c] = sum of y over rows [64p,64p+64) ; stats[1][p][c] = sum of (y - partial mean)^2.
 * data-grad = the same kernel on dy with the packed weights of cvk_pack_weight_dgrad. */
int cvk_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, float* stats,
                    int N, int H, int W, int Cin, int Cout, int ldy, void* stream);
/* w_src: [Cout][9][Cin] -> dst [Cout][9][Cin_pad] (zero padded). */
int cvk_pack_weight_fwd(const float* w_src, float* dst, int Cout, int Cin, int Cin_pad, void* stream);
/* w_src: [Cout][9][Cin] -> dst [Cin_pad][9][Cout_pad], dst[ci][t][co] = w_src[co][8-t][ci] (flipped taps). */
int cvk_pack_weight_dgrad(const float* w_src, float* dst, int Cout, int Cin, int Cin_pad, int Cout_pad, void* stream);
/* weight-grad (ConvolutionBackward of train.py:131): dw[co][tap][ci] = sum_m dy[m][co] * x[m+tap][ci].
 *   x: dense, ld = Cin_pad; dy: dense, ld = ld_dy; dw: [Cout][9][Cin] (Cin <= Cin_pad real channels).
 *   Split over pixels into fp32 slabs in `workspace`, then a deterministic slab reduction. */
size_t cvk_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout);
int cvk_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                      int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream);

/* Same operator through 1-D Winograd F(2,3) along the width (1.5x fewer MFMA FLOPs; csrc/wino.hip), for Cin % 64 == 0:
 *   U  = cvk_wino_weight_transform(w)                      [4][Cout][3][Cin] from w [Cout][3][3][Cin]
 *   Mo = cvk_conv3x3_wino_gemm(x, U)                       four transformed products, float[4][N*H*ceil(W/2)][ldm]
 *                                                          (cvk_conv3x3_wino_workspace_bytes(N,H,W,ldm) bytes)
 *   y, stats = cvk_wino_output(Mo, bias)                   output transform + bias + fused BN statistics partials;
 *                                                          y / stats / bias / ldy exactly as in cvk_conv3x3_fwd (ldy == ldm)
 * Data-grad: the same three calls on dy with the cvk_pack_weight_dgrad weights, bias = stats = NULL. */
int cvk_wino_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream);
size_t cvk_conv3x3_wino_workspace_bytes(int N, int H, int W, int Cout_ld);
int cvk_conv3x3_wino_gemm(const float* x, const float* U, float* Mo, int N, int H, int W, int Cin, int Cout, int ldm,
                          void* stream);
int cvk_wino_output(const float* Mo, const float* bias, float* y, float* stats, int N, int H, int W, int Cout, int ldy,
                    void* stream);

/* The same three steps through 1-D Winograd F(4,3) (2x fewer MFMA FLOPs than the direct form; csrc/wino4.hip):
 *   U [6][Cout][3][Cin],  Mo float[ksplit][6][N*H*ceil(W/4)][ldm],  otherwise the contract of the F(2,3) calls above.
 * Layers with few tiles split the K = 3*Cin reduction over ksplit = cvk_conv3x3_wino4_ksplit(...) in {1,2,3} workgroups per
 * transform index (a fixed function of the shape); the partial planes are summed by cvk_wino4_output, which takes that
 * ksplit, and the workspace size query accounts for them. */
int cvk_wino4_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream);
/* data-grad weights in one step: U[6][Cin][3][Cout] of the rotated, channel-transposed filter, straight from
 * w [Cout][3][3][Cin] (== cvk_pack_weight_dgrad without padding followed by cvk_wino4_weight_transform) */
int cvk_wino4_weight_transform_dgrad(const float* w, float* U, int Cout, int Cin, void* stream);
int cvk_conv3x3_wino4_ksplit(int N, int H, int W, int Cin, int Cout_ld);
size_t cvk_conv3x3_wino4_workspace_bytes(int N, int H, int W, int Cin, int Cout_ld);
int cvk_conv3x3_wino4_gemm(const float* x, const float* U, float* Mo, int N, int H, int W, int Cin, int Cout, int ldm,
                           void* stream);
int cvk_wino4_output(const float* Mo, const float* bias, float* y, float* stats, int N, int H, int W, int Cout, int ldy,
                     int ksplit, void* stream);

/* 2-D Winograd F(4x4,3x3) for channel-heavy layers (csrc/wino2d.hip): forward / data-grad of nn.Conv2d(k=3,pad=1)
 * (reference models/unet.py:11, bwd of train.py:131) as 36 batched GEMMs over T = cvk_w2d_tiles(N,H,W) 4x4 output tiles.
 *   U = cvk_w2d_weight_transform(w)        float[36][Cout][Cin] from w [Cout][3][3][Cin] (data-grad: from the
 *                                          cvk_pack_weight_dgrad pack, roles of Cin / Cout exchanged)
 *   cvk_conv3x3_w2d(x, U, ...)             input transform -> GEMMs -> output transform + bias; x dense [N,H,W,Cin],
 *                                          Cin % 32 == 0, Cout % 4 == 0, Cout >= 64; workspace of
 *                                          cvk_conv3x3_w2d_workspace_bytes(N,H,W,Cin,Cout) bytes.
 *   stats/counts (both or neither): BatchNorm statistics partials float[2][P][Cout] (sum, M2 about the partial mean) and
 *   float[P] pixel counts, P = cvk_w2d_stat_partials(N,H,W) -> cvk_bn_finalize_counts. */
int cvk_w2d_tiles(int N, int H, int W);
int cvk_w2d_tpad(int T);                      /* rows per transform plane: T rounded up to a multiple of 32 (zero rows) */
int cvk_w2d_stat_partials(int N, int H, int W);
size_t cvk_conv3x3_w2d_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int cvk_w2d_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream);
/* the data-grad filter float[36][Cin][Cout] straight from the FORWARD weights w [Cout][3][3][Cin] (rotation by 180 degrees and
 * the channel exchange happen inside; equals cvk_w2d_weight_transform(cvk_pack_weight_dgrad(w)) bit for bit) */
int cvk_w2d_weight_transform_dgrad(const float* w, float* U, int Cout, int Cin, void* stream);
int cvk_conv3x3_w2d(const float* x, const float* U, const float* bias, float* y, float* stats, float* counts, int N, int H,
                    int W, int Cin, int Cout, int ldy, void* workspace, size_t workspace_bytes, void* stream);
/* the three passes of cvk_conv3x3_w2d, callable (and timed) one by one.  V float[36][Tpad][Cin] + 512 bytes of slack,
 * Tpad = cvk_w2d_tpad(T), rows beyond T zero; Mo float[f][36][T][Cout] with f = cvk_w2d_ksplit(T, Cin, Cout) K-range planes
 * (the tiles of the last partial round of workgroups are split in K; the output pass adds the planes in a fixed order and
 * therefore takes Cin as well) */
int cvk_w2d_ksplit(int T, int Cin, int Cout);
int cvk_w2d_input_transform(const float* x, float* V, int N, int H, int W, int Cin, void* stream);
int cvk_w2d_gemm(const float* V, const float* U, float* Mo, int T, int Cin, int Cout, void* stream);
int cvk_w2d_output(const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H, int W, int Cin,
                   int Cout, int ldy, void* stream);
/* weight-grad through the transposed 2-D F(4x4,3x3) (contract of cvk_conv3x3_wgrad; channel counts multiples of 4):
 * dW = G^T [ sum_tiles (A dy A^T) (.) (B^T x B) ] G, 36 GEMMs whose depth is the tile index.  x == NULL: the first
 * 36*Tpad*Cin_pad floats of the workspace already hold V = cvk_w2d_input_transform(x) (kept from the forward pass).
 * Passes: E float[36][Tpad][Cout] (+ slack) = cvk_w2d_dy_transform(dy); P float[f][36][Cout][Cin_pad] = cvk_w2d_gemm_tn(E, V),
 * f = cvk_w2d_wgrad_ksplit(T, Cin_pad, Cout) depth ranges; dw = cvk_w2d_wgrad_output(P) adds them in a fixed order. */
size_t cvk_conv3x3_wgrad_w2d_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout);
int cvk_conv3x3_wgrad_w2d(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout,
                          int ld_dy, void* workspace, size_t workspace_bytes, void* stream);
int cvk_w2d_wgrad_ksplit(int T, int Cin_pad, int Cout);
int cvk_w2d_dy_transform(const float* dy, int ld_dy, float* E, int N, int H, int W, int Cout, void* stream);
/* both transforms of dy in one launch: Vp = cvk_w2d_input_transform(dy) (the data-grad's operand, dy dense with row stride ld_dy =
 * Cout required by that GEMM) and E = cvk_w2d_dy_transform(dy) (the weight-grad's), bit for bit; dy crosses the fabric once */
int cvk_w2d_dy_transform_both(const float* dy, int ld_dy, float* Vp, float* E, int N, int H, int W, int Cout, void* stream);
int cvk_w2d_gemm_tn(const float* E, const float* V, float* P, int T, int Cin_pad, int Cout, void* stream);
int cvk_w2d_wgrad_output(const float* P, float* dw, int T, int Cin, int Cin_pad, int Cout, void* stream);

/* 2-D Winograd F(6x6,3x3) (csrc/wino2d.hip, the same kernels instantiated for 8x8 input tiles; points 0, +-1, +-2, +-1/2, inf):
 * 64 batched GEMMs over T = cvk_w6_tiles(N,H,W) 6x6 output tiles — 1.78 multiplies per output and input channel instead of
 * 2.25, and 1.78x instead of 2.25x the activation in transform-domain planes.  fp32 rounding about twice that of F(4x4,3x3)
 * (5e-6 relative L2 at 256 input channels).  Every cvk_w6_* entry point has the contract of its cvk_w2d_* namesake with 36 -> 64
 * planes: U float[64][Cout][Cin], V float[64][Tpad][Cin] (+ 512 bytes), Mo float[f][64][T][Cout], E float[64][Tpad][Cout],
 * P float[f][64][Cout][Cin_pad]; Tpad = cvk_w2d_tpad(T). */
int cvk_w6_tiles(int N, int H, int W);
int cvk_w6_stat_partials(int N, int H, int W);
size_t cvk_conv3x3_w6_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int cvk_w6_weight_transform(const float* w, float* U, int Cout, int Cin, void* stream);
int cvk_w6_weight_transform_dgrad(const float* w, float* U, int Cout, int Cin, void* stream);
int cvk_w6_ksplit(int T, int Cin, int Cout);
int cvk_w6_input_transform(const float* x, float* V, int N, int H, int W, int Cin, void* stream);
int cvk_w6_gemm(const float* V, const float* U, float* Mo, int T, int Cin, int Cout, void* stream);
int cvk_w6_output(const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H, int W, int Cin,
                  int Cout, int ldy, void* stream);
int cvk_w6_wgrad_ksplit(int T, int Cin_pad, int Cout);
int cvk_w6_dy_transform(const float* dy, int ld_dy, float* E, int N, int H, int W, int Cout, void* stream);
int cvk_w6_dy_transform_both(const float* dy, int ld_dy, float* Vp, float* E, int N, int H, int W, int Cout, void* stream);
int cvk_w6_gemm_tn(const float* E, const float* V, float* P, int T, int Cin_pad, int Cout, void* stream);
/* OPT-IN SPLIT-OPERAND PATH (round 5; the executor's default fp32 path is exact-fp32 MFMA — this one runs only under runner.w2d_split): the
 * GEMM stage above on the 16-bit matrix pipe with split fp32 operands, accumulated in fp32 (csrc/split3.hip, csrc/split_fmt.h, tools/study/).
 *   fmt 3: x = x1 + x2 + x3 in bf16, the six largest cross-products (v_mfma_f32_16x16x32_bf16): fp32-MFMA accuracy at 0.375 of its matrix time;
 *   fmt 2: 2^e * x = h1 + h2 in fp16, three cross-products (v_mfma_f32_16x16x32_f16): 0.19 of the matrix time.  The power-of-two scale per
 *          transform index comes from the EXACT largest magnitude of the tensor a transform reads and the absolute row sums of the transform
 *          matrix, so no element can leave fp16's range whatever the data; the GEMMs undo the scale exactly.  The magnitude lives in an "amax
 *          block" of device memory: cvk_amax_block_words() 32-bit words (8 slots one cache line apart, each an atomicMax of fp32 bit patterns —
 *          one hot word would serialise tens of thousands of waves in the L2; the value is the maximum over the slots, cvk_amax_block_value on a
 *          host copy).  Zero the block, then cvk_absmax_f32 (a pass over the tensor) or the *_amax variants of the passes that WRITE the tensor
 *          (below) fill it; every `amax*` argument of this section is such a block.  cvk_split_scale_exponent (host) returns e for (tile, kind, xi, word):
 *          kind 0 = B (input transforms V, V'), 1 = G (filters U), 2 = A (dy -> E).
 * Split planes: 16-bit [NX][C/32][fmt][Rpad][32] (Rpad = cvk_split3_rows_pad(R, 256) for V / V' / E, (R, 128) for U; C % 32 == 0).
 * cvk_split_planes / cvk_split3_planes: fp32 planes P[NX][R][C] -> split planes (a stand-alone pass: tests and studies).
 * cvk_w2d_gemm_split / _split3: Mo fp32 [NX][T][Cout] = V * U^T per transform index, both operands as split planes.  The *_split3 names are
 * the fmt-3 forms of the generic entry points (no amax words). */
int cvk_split3_rows_pad(int R, int mult);
int cvk_split3_planes(const float* P, void* S, int NX, int R, int Rpad, int C, void* stream);
int cvk_w2d_gemm_split3(const void* V3, const void* U3, float* Mo, int NX, int T, int Tpad, int Cin, int Cout, int Cpad, void* stream);
int cvk_amax_block_words(void);
unsigned cvk_amax_block_value(const unsigned* host_words, int n);
int cvk_absmax_f32(const float* x, long rows, int C, int ld, void* amax_block, void* stream);
int cvk_split_scale_exponent(int tile, int kind, int xi, unsigned amax_block);
int cvk_split_planes(int fmt, int tile, int kind, const float* P, void* S, const void* amax, int NX, int R, int Rpad, int C, void* stream);
int cvk_w2d_gemm_split(int fmt, int tile, const void* V, const void* U, float* Mo, const void* amax_v, const void* amax_u, int NX, int T, int Tpad,
                       int Cin, int Cout, int Cpad, void* stream);
/* ... the rest of the split-operand path (tile = 4 or 6 selects F(4x4,3x3) / F(6x6,3x3)): transforms that write split planes from their store loops
 * (V3 / Vp3 / E3 rows padded to cvk_split3_rows_pad(T, 256); E stays fp32 [NX][cvk_w2d_tpad(T)][C] with e_split = 0), the filter as split
 * planes (tmp: NX * Cout * Cin floats), the output pass for product planes without K-range partials, and the weight-grad GEMM
 * P[f][NX][Cout][Cin] = E^T V on split planes (f = cvk_w2d_gemm_tn_split3_ksplit; Cout % 256 == 0 && Cin % 128 == 0 or the reverse) with
 * its final pass for an explicit f. */
int cvk_w2d_input_transform_split3(int tile, const float* x, void* V3, int N, int H, int W, int Cin, void* stream);
int cvk_w2d_dy_transform_both_split3(int tile, const float* dy, int ld_dy, void* Vp3, void* E, int e_split, int N, int H, int W, int C,
                                     void* stream);
int cvk_w2d_weight_transform_split3(int tile, const float* w, void* U3, float* tmp, int Cout, int Cin, int dgrad, void* stream);
int cvk_w2d_input_transform_split(int fmt, int tile, const float* x, void* V, const void* amax_x, int N, int H, int W, int Cin, void* stream);
int cvk_w2d_dy_transform_both_split(int fmt, int tile, const float* dy, int ld_dy, void* Vp, void* E, int e_split, const void* amax_dy, int N, int H,
                                    int W, int C, void* stream);
int cvk_w2d_weight_transform_split(int fmt, int tile, const float* w, void* U, const void* amax_w, int Cout, int Cin, int dgrad, void* stream);
int cvk_w2d_output_plain(int tile, const float* Mo, const float* bias, float* y, float* stats, float* counts, int N, int H, int W,
                         int Cout, int ldy, void* stream);
int cvk_w2d_gemm_tn_split3_ksplit(int NX, int Tpad, int Cin, int Cout);
int cvk_w2d_gemm_tn_split3(const void* E3, const void* V3, float* P, int NX, int Tpad, int Cin, int Cout, void* stream);
int cvk_w2d_gemm_tn_split(int fmt, int tile, const void* E, const void* V, float* P, const void* amax_e, const void* amax_v, int NX, int Tpad, int Cin,
                          int Cout, void* stream);
int cvk_w2d_wgrad_output_f(int tile, const float* P, float* dw, int Cin, int Cin_pad, int Cout, int f, void* stream);
int cvk_w6_wgrad_output(const float* P, float* dw, int T, int Cin, int Cin_pad, int Cout, void* stream);
/* The fused F(4,3) convolution below in the opt-in fp16 split-operand form (csrc/split_fmt.h; runner.w2d_split = 2, never the default): same
 * contracts as cvk_wino4f_weight_transform / cvk_conv3x3_wino4f / cvk_conv3x3_wino4f_bnred plus the amax blocks of x and of the filter w; Uh has the
 * size of Uf (cvk_wino4f_weight_floats(Cn, Ck) * 4 bytes).  V = B^T d is split into two fp16 terms inside the staging pass, a K step is 18
 * v_mfma_f32_32x32x16_f16 instead of 48 v_mfma_f32_32x32x2f32. */
int cvk_wino4h_weight_transform(const float* w, void* Uh, const void* amax_w, int Cn, int Ck, int dgrad, void* stream);
int cvk_conv3x3_wino4h(const float* x, const void* Uh, const float* bias, float* y, float* stats, float* counts, const void* amax_x,
                       const void* amax_w, int N, int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream);
int cvk_conv3x3_wino4h_bnred(const float* x, const void* Uh, float* y, const void* amax_x, const void* amax_w, int N, int H, int W, int Cin,
                             int Cout, int ldy, const float* yP, const float* scale, const float* shift, const float* mean,
                             const float* rstd, float* part, int max_workgroups, void* stream);
/* FUSED 1-D Winograd F(4,3) (csrc/wino4f.hip; replaces nn.Conv2d(cin,cout,3,padding=1) fwd and its data-grad,
 * /root/reference/models/unet.py:11, models/segnet.py:8, for the 64/128-channel levels): one workgroup computes all six
 * transform indices of a 128 x 64 tile, the output transform, bias and BatchNorm statistics happen in registers — no
 * product planes, no output pass.  Uf = cvk_wino4f_weight_transform(w): cvk_wino4f_weight_floats(Cn, Ck) floats, the
 * K-sliced LDS image of (G g); dgrad != 0 builds the data-grad filter (180-degree rotation, channels exchanged) straight
 * from the forward weights w [Ck][3][3][Cn].  Cin % 32 == 0.  stats/counts (both or neither): P =
 * cvk_wino4f_stat_partials(N,H,W) partials [sum | M2 about the partial mean] [2][P][Cout] + the P pixel counts, to be
 * reduced by cvk_bn_finalize_counts.  The kernel is persistent (one workgroup per CU walks a list of tiles);
 * max_workgroups > 0 caps its grid (data-parallel runs leave CUs to RCCL's kernels), 0 = one per CU. */
size_t cvk_wino4f_weight_floats(int Cn, int Ck);
int cvk_wino4f_weight_transform(const float* w, float* Uf, int Cn, int Ck, int dgrad, void* stream);
/* n (<= CVK_WT_BATCH_MAX) filter transforms in ONE launch; `jobs` is a HOST array (copied into the kernel arguments).  A training step
 * rebuilds every Winograd-domain filter (the weights changed in optimizer.step(), train.py:134): 16 + 26 launches of 5-15 us each.
 *   cvk_wino4f_weight_transform_batch: job = cvk_wino4f_weight_transform(w, out, rows = Cn, cols = Ck, dgrad)      (tile ignored)
 *   cvk_w2d_weight_transform_batch:    job = cvk_w2d_ / cvk_w6_weight_transform[_dgrad](w, out, rows = Cout, cols = Cin), tile = 4 | 6 */
#define CVK_WT_BATCH_MAX 48
typedef struct cvk_wt_job { const float* w; float* out; int rows, cols, tile, dgrad; } cvk_wt_job;
int cvk_wino4f_weight_transform_batch(const cvk_wt_job* jobs, int n, void* stream);
int cvk_w2d_weight_transform_batch(const cvk_wt_job* jobs, int n, void* stream);
int cvk_wino4f_stat_partials(int N, int H, int W);
/* cvk_conv3x3_wino4f (forward form) that ALSO leaves the layer's weight-grad operand behind: the kernel's staging path computes exactly the
 * rows V_xi = B^T d that the plane GEMM of the weight-grad reads (cvk_wgradp_gemm_sm), so the centre-kernel-row slices are stored as six
 * SLICE-MAJOR planes Vsm[6][Cin / 16][cvk_wgradp_plane_rows(N,H,W)][16] (element (xi, row, c) at ((xi * Cin/16 + c/16) * rows + row) * 16 + c % 16;
 * a wave's store is 1 KiB contiguous).  The pad rows must have been zeroed by cvk_wgradp_zero_pads_sm.  y / stats / counts are bitwise those of
 * cvk_conv3x3_wino4f; stats and counts may both be NULL (frozen BatchNorm).  Replaces the x -> V pass of the weight-grad of
 * nn.Conv2d(cin,cout,3,padding=1) (/root/reference/models/unet.py:11, backward of train.py:131). */
int cvk_conv3x3_wino4f_vplanes(const float* x, const float* Uf, const float* bias, float* y, float* stats, float* counts, float* Vsm,
                               int N, int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream);
int cvk_conv3x3_wino4f(const float* x, const float* Uf, const float* bias, float* y, float* stats, float* counts, int N, int H,
                       int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream);
/* The data-grad launch whose result y is the COMPLETE dL/d(activation) of the block that produced this conv's input
 * (nn.BatchNorm2d + ReLU backward of /root/reference/models/unet.py:12-13 starts with two column sums over exactly that
 * tensor): the epilogue also leaves  sum g  and  sum g * xhat,  g = y masked by the producer's ReLU (yP*scale+shift > 0),
 * xhat = (yP - mean) * rstd,  as part = float[2][cvk_wino4f_stat_partials(N,H,W)][Cout] for
 * cvk_colsum_finalize(part, partials, Cout, dbeta, dgamma) — cvk_bn_bwd_reduce is then not launched for that block.
 * yP: the producer's conv output [N*H*W][ldy] (same row stride as y); no bias, no forward statistics. */
int cvk_conv3x3_wino4f_bnred(const float* x, const float* Uf, float* y, int N, int H, int W, int Cin, int Cout, int ldy,
                             const float* yP, const float* scale, const float* shift, const float* mean, const float* rstd,
                             float* part, int max_workgroups, void* stream);

/* THIN layers (csrc/thin.hip): the stem nn.Conv2d(3, 64, 3, padding=1) (/root/reference/models/unet.py:103,
 * models/segnet.py first block) and the classifier head nn.Conv2d(64, class_num, 3, padding=1) (models/unet.py:127), forward,
 * data-grad and weight-grad.  The thin channel dimension is one 16-row side (or the k = 4) of v_mfma_f32_16x16x4_f32; a wave
 * walks down a 16-pixel (forward) / 4-pixel (weight-grad) column with the 3x3 window's rows in registers.
 *   cvk_conv3x3_thin_fwd: y = conv3x3(x, w) + bias; w [Cout][9][Cin_ld].  stats/counts (both or neither):
 *     P = cvk_thin_stat_partials(N,H,W,Cin_ld) partials [sum | M2 about the partial mean] [2][P][Cout] + P pixel counts for
 *     cvk_bn_finalize_counts.  Supported shapes: cvk_thin_fwd_supported(Cin_ld, Cout, ldy) != 0.
 *   cvk_conv3x3_thin_wgrad: dw [Cout][9][Cin] = sum_p dy[p][co] * x[p + tap][ci]; fixed-order reduction (bitwise
 *     reproducible).  Supported shapes: cvk_thin_wgrad_supported(Cin, Cin_ld, Cout, ld_dy) != 0. */
int cvk_thin_fwd_supported(int Cin_ld, int Cout, int ldy);
int cvk_thin_stat_partials(int N, int H, int W, int Cin_ld);
int cvk_conv3x3_thin_fwd(const float* x, const float* w, const float* bias, float* y, float* stats, float* counts, int N, int H,
                         int W, int Cin_ld, int Cout, int ldy, void* stream);
int cvk_thin_wgrad_supported(int Cin, int Cin_ld, int Cout, int ld_dy);
size_t cvk_conv3x3_thin_wgrad_workspace_bytes(int N, int H, int W, int Cin_ld, int Cout);
int cvk_conv3x3_thin_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_ld, int Cout,
                           int ld_dy, void* workspace, size_t workspace_bytes, void* stream);

/* Transposed F(4,3) weight-grad with both transforms outside the GEMM (csrc/wgradp.hip; the weight gradient of
 * nn.Conv2d(cin,cout,3,padding=1), /root/reference/models/unet.py:11, backward of train.py:131): x and dy are written once as
 * transform-domain planes (1.5x each, zero rows at every image border), the GEMM is one wave per workgroup with no vector
 * arithmetic and no barrier.  Contract of cvk_conv3x3_wgrad; Cin_pad % 64 == 0 and Cout % 64 == 0. */
size_t cvk_conv3x3_wgradp_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout);
int cvk_conv3x3_wgradp(const float* x, const float* dy, const float* E6_pre, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                       int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream);
/* E6_pre: NULL (the E planes are built from dy inside the call) or six planes float[6][cvk_wgradp_plane_rows(N,H,W)][Cout] that
 * the BatchNorm/ReLU-backward pass wrote on its way: cvk_wgradp_zero_pads (pad rows) then cvk_bn_bwd_dx_e6 (contract of
 * cvk_bn_bwd_dx_e with six planes E0..E5 in the padded plane layout; ld_dy == C). */
long cvk_wgradp_plane_rows(int N, int H, int W);
/* the steps of cvk_conv3x3_wgradp, separately callable: planes of x (is_dy == 0) or dy (is_dy != 0), then GEMM + reduction */
int cvk_wgradp_planes(const float* t, int ld, float* planes, int N, int H, int W, int C, int is_dy, void* stream);
size_t cvk_wgradp_gemm_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout);
int cvk_wgradp_gemm(const float* E6, const float* V6, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout, void* workspace,
                    size_t workspace_bytes, void* stream);
int cvk_wgradp_zero_pads(float* planes, int N, int H, int W, int C, void* stream);
/* the same three steps for SLICE-MAJOR V planes [6][C / 16][rows][16] (C % 16 == 0) — the layout cvk_conv3x3_wino4f_vplanes writes from the forward
 * pass, so that the weight-grad needs no pass over x at all: zero the pad rows before the forward launch; cvk_wgradp_planes_sm builds the
 * same planes from x in a pass of its own (forward ran another kernel; tests); cvk_wgradp_gemm_sm = cvk_wgradp_gemm reading them (E6 stays
 * row-major [6][rows][Cout]; same workspace size). */
int cvk_wgradp_zero_pads_sm(float* planes, int N, int H, int W, int C, void* stream);
int cvk_wgradp_planes_sm(const float* x, int ld, float* planes, int N, int H, int W, int C, void* stream);
int cvk_wgradp_gemm_sm(const float* E6, const float* V6sm, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout, void* workspace,
                       size_t workspace_bytes, void* stream);
/* ... with only FOUR E planes: E0 and E5 of E = A dy are columns 4 xt and 4 xt + 3 of dy itself, so the BatchNorm-backward pass writes E1..E4 only
 * (cvk_wgradp_zero_pads4 for the pad rows, then cvk_bn_bwd_dx_e4p: contract of cvk_bn_bwd_dx_e6 with float E4p[4][cvk_wgradp_plane_rows][C]) and the GEMM
 * reads the two identity planes from dy: dense [N*H*W][Cout] followed by cvk_wgradp_dy_slack(W) * Cout ZERO floats (pad column groups of the last row
 * read past the tensor; their V rows are zero).  W % 4 == 0 (a ragged last group's E5 is 0, not a pixel of dy).  1.0x instead of 1.5x the tensor
 * written beside dy. */
int cvk_wgradp_zero_pads4(float* planes, int N, int H, int W, int C, void* stream);
int cvk_bn_bwd_dx_e4p(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                      const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E4p, float* part,
                      int N, int H, int W, int C, int use_batch_stats, void* stream);
int cvk_wgradp_dy_slack(int W);
int cvk_wgradp_gemm_sm_dy(const float* E4p, const float* dy, const float* V6sm, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout,
                          void* workspace, size_t workspace_bytes, void* stream);
int cvk_bn_bwd_dx_e6(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                     const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E6, float* part,
                     int N, int H, int W, int C, int use_batch_stats, void* stream);

/* weight-grad through the transposed F(4,3) (contract of cvk_conv3x3_wgrad; the workspace also holds the transformed
 * output-gradient planes E1..E4, float[4][N*H*ceil(W/4)][ld_dy], hence the extra ld_dy argument of the size query) */
size_t cvk_conv3x3_wgrad_wino4_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout, int ld_dy);
int cvk_conv3x3_wgrad_wino4(const float* x, const float* dy, const float* E_pre, float* dw, int N, int H, int W, int Cin,
                            int Cin_pad, int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream);
/* E_pre: NULL (the call transforms dy itself) or the planes written by cvk_bn_bwd_dx_e — the BN/ReLU-backward pass that
 * produces dy can emit them on the way (same contract as cvk_bn_bwd_dx plus E; 4-channel vector layout only, CVK_EINVAL
 * otherwise; `part` gets cvk_bn_bwd_e_blocks(N,H,W) partial rows of C column sums). */
int cvk_bn_bwd_e_blocks(int N, int H, int W);
int cvk_bn_bwd_dx_e(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                    const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E, float* part,
                    int N, int H, int W, int C, int use_batch_stats, void* stream);

/* weight-grad through the transposed F(2,3) (same contract as cvk_conv3x3_wgrad; any Cin_pad % 4 == 0) */
size_t cvk_conv3x3_wgrad_wino_workspace_bytes(int N, int H, int W, int Cin_pad, int Cout);
int cvk_conv3x3_wgrad_wino(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cin_pad,
                           int Cout, int ld_dy, void* workspace, size_t workspace_bytes, void* stream);

/* ---- BatchNorm2d (+ReLU) (models/unet.py:12-13, models/segnet.py:9-10) -------------------------------------------
 * finalize (training): combines the conv-epilogue partials (Chan's parallel variance, fp64) into per-channel
 *   mean / rstd = 1/sqrt(biased var + eps), scale = gamma*rstd, shift = beta - mean*scale, and updates
 *   running_mean/var (unbiased var, momentum) and num_batches_tracked in place when those pointers are non-NULL. */
size_t cvk_bn_finalize_workspace_bytes(int P, int C);
int cvk_bn_finalize(const float* stats, int P, int M, int C, const float* gamma, const float* beta,
                    float* mean, float* rstd, float* scale, float* shift,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, void* workspace, size_t workspace_bytes, void* stream);
/* eval mode: scale/shift from running statistics (train.py:169 net.eval()); also fills mean/rstd. */
int cvk_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float* mean, float* rstd, float* scale, float* shift, int C, float eps, void* stream);
/* out = max(0, y*scale + shift), y dense (ldy), out a view. */
int cvk_bn_relu_apply(const float* y, int ldy, const float* scale, const float* shift, cvk_view out,
                      int N, int H, int W, int C, void* stream);
/* the same with nn.MaxPool2d(2,2) of the result fused in (/root/reference/models/unet.py:100-109, models/segnet.py:79): pool
 * float[N][H/2][W/2][C] dense, code (optional) uint8 arg-max codes as cvk_maxpool2x2_fwd; 4-channel vector layout only */
int cvk_bn_relu_apply_pool(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, float* pool,
                           unsigned char* code, int N, int H, int W, int C, void* stream);
/* backward, pass 1: g = dout * [y*scale+shift > 0]; partial sums of g and g*xhat per row block:
 *   part float[2][PB][C], PB = cvk_bn_bwd_blocks(M). */
int cvk_bn_bwd_blocks(int M);
int cvk_bn_bwd_reduce(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift,
                      const float* mean, const float* rstd, float* part, int N, int H, int W, int C, void* stream);
/* sums PB partial rows in fp64: out0[c] = sum part[0][:,c], out1[c] = sum part[1][:,c] (out1/part1 nullable) */
int cvk_colsum_finalize(const float* part, int PB, int C, float* out0, float* out1, void* stream);
/* n (<= CVK_COLSUM_BATCH_MAX) such finalisations, one output each, in ONE launch; `jobs` is a HOST array (copied into the
 * kernel arguments).  The executor collects the conv-bias gradients of a backward pass and finalises them at its end
 * (or when a data-parallel gradient bucket is handed to the all-reduce). */
#define CVK_COLSUM_BATCH_MAX 64
typedef struct cvk_colsum_job { const float* part; float* out; int PB, C; } cvk_colsum_job;
int cvk_colsum_finalize_batch(const cvk_colsum_job* jobs, int n, void* stream);
/* backward, pass 2: dy = scale*(g - dbeta/M - xhat*dgamma/M) (training) or dy = scale*g (eval: use_batch_stats=0);
 *   also emits column-sum partials of dy (conv bias gradient) into dbias_part float[PB][C] when non-NULL. */
/* Variants that also leave the EXACT largest magnitude they write in a device word (atomicMax of fp32 bit patterns; the caller zeroes the word
 * first) — what cvk_absmax_f32 of the written tensor would return, without the extra pass.  The opt-in fp16 split-operand path
 * (cvk_w2d_*_split with fmt 2) scales its planes by these words.  cvk_bn_relu_apply_pool_amax: either word may be NULL. */
int cvk_bn_relu_apply_amax(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, int N, int H, int W, int C,
                           void* amax_block, void* stream);
int cvk_bn_relu_apply_pool_amax(const float* y, int ldy, const float* scale, const float* shift, cvk_view out, float* pool,
                                unsigned char* code, int N, int H, int W, int C, void* amax_out, void* amax_pool, void* stream);
int cvk_bn_bwd_dx_e_amax(int six, cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                         const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* E, float* part, int N, int H,
                         int W, int C, int use_batch_stats, void* amax_block, void* stream);   /* six: 0 = cvk_bn_bwd_dx_e, 1 = cvk_bn_bwd_dx_e6 */
int cvk_bn_bwd_dx_amax(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift, const float* mean,
                       const float* rstd, const float* dgamma, const float* dbeta, float* dy, int ld_dy, float* dbias_part, int N,
                       int H, int W, int C, int use_batch_stats, void* amax_block, void* stream);
int cvk_bn_bwd_dx(cvk_view dout, const float* y, int ldy, const float* scale, const float* shift,
                  const float* mean, const float* rstd, const float* dgamma, const float* dbeta,
                  float* dy, int ld_dy, float* dbias_part, int N, int H, int W, int C, int use_batch_stats, void* stream);

/* ---- MaxPool2d(2,2) (models/unet.py:92; with indices models/segnet.py:79) and MaxUnpool2d (segnet.py:80) ------
 * floor output size; first maximum wins ties.  code (nullable): uint8 window position 0..3 (= 2*dy+dx) of the
 * arg-max, the compact form of torch's int64 flat indices. */
int cvk_maxpool2x2_fwd(cvk_view x, float* out, uint8_t* code, int N, int H, int W, int C, void* stream);
/* dx(view) (+)= route(dout) ; arg-max recomputed from x when code == NULL.  accumulate=0 overwrites (and zeroes
 * the odd trailing row/column), accumulate=1 adds. */
int cvk_maxpool2x2_bwd(const float* dout, cvk_view x, const uint8_t* code, cvk_view dx, int accumulate,
                       int N, int H, int W, int C, void* stream);
/* ... that also leaves the first pass of the PRODUCING block's BatchNorm+ReLU backward (nn.BatchNorm2d + nn.ReLU of models/unet.py:12-13 in front of the
 * nn.MaxPool2d of :92; backward of train.py:131): the pool backward is the last writer of the block's output gradient dx and touches every element, so it
 * sums g = dx * [ReLU passed] and g * xhat on the way — part = float[2][cvk_maxpool2x2_bwd_bnred_blocks(N,H,W,C)][C] for cvk_colsum_finalize; yP [N*H*W][ldp]
 * is the block's conv output, scale / shift / mean / rstd as for cvk_bn_bwd_reduce, which is then not launched.  _blocks returns 0 for unsupported C. */
int cvk_maxpool2x2_bwd_bnred_blocks(int N, int H, int W, int C);
int cvk_maxpool2x2_bwd_bnred(const float* dout, cvk_view x, const uint8_t* code, cvk_view dx, int accumulate, int N, int H, int W, int C,
                             const float* yP, int ldp, const float* scale, const float* shift, const float* mean, const float* rstd,
                             float* part, void* stream);
int cvk_maxunpool2x2_fwd(const float* v, const uint8_t* code, float* out, int N, int H, int W, int C, void* stream);
int cvk_maxunpool2x2_bwd(const float* dout, const uint8_t* code, float* dv, int N, int H, int W, int C, void* stream);
/* uint8 codes -> torch-style int64 flat H*W indices in NCHW order (API parity with return_indices=True) */
int cvk_pool_code_to_index(const uint8_t* code, int64_t* idx, int N, int H, int W, int C, void* stream);

/* ---- bilinear x2, align_corners=True (nn.Upsample: models/unet.py:25) -----------------------------------------
 * x: dense [N,H,W,C] (ld=C) -> out dense [N,2H,2W,C];  bwd: dx = transpose(dout). */
int cvk_bilinear_up2_fwd(const float* x, float* out, int N, int H, int W, int C, void* stream);
int cvk_bilinear_up2_bwd(const float* dout, float* dx, int N, int H, int W, int C, void* stream);

/* ---- softmax cross-entropy, mean over pixels (nn.CrossEntropyLoss(): train.py:105,130-131) --------------------
 * logits: dense NHWC rows [M][ld] (ld <= 128), target int64 [M].  Pixels whose target equals ignore_index (torch's
 * default -100; the reference passes none, so every CamVid class incl. Void is trained) are left out of the mean and get
 * a zero gradient.  fwd writes loss[0] = mean over the valid pixels, loss[1] = their number, loss[2] = number of
 * targets outside [0,C) that are not ignore_index (then loss[0] = NaN: torch raises there); `part` is scratch of
 * 3 * cvk_ce_blocks(M) floats.  bwd: dlogits = (softmax - onehot) * (*grad_out) * scale / loss3[1]; loss3 is fwd's loss. */
int cvk_ce_blocks(int M);
int cvk_softmax_ce_fwd(const float* logits, int ld, const int64_t* target, float* part, float* loss,
                       int M, int C, int ignore_index, void* stream);
int cvk_softmax_ce_bwd(const float* logits, int ld, const int64_t* target, const float* loss3, const float* grad_out,
                       float scale, float* dlogits, int ld_d, int M, int C, int ignore_index, void* stream);

/* ---- evaluation (train.py:191 argmax; utils.py:162-190 histograms) -------------------------------------------- */
int cvk_argmax_channels(const float* logits, int ld, int64_t* out, int M, int C, void* stream);
/* hist int64[3][num_classes] += (intersection, prediction area, label area), pixels with label==ignore skipped */
int cvk_confusion_accumulate(const int64_t* pred, const int64_t* label, int64_t* hist, int M, int num_classes,
                             int ignore_index, void* stream);

/* ---- input pipeline on device (transforms.ToTensor + Normalize: transforms.py:485-538, MEAN/STD conf/settings.py:8-9) ----
 * src uint8 [N,H,W,3] (channel order as decoded, i.e. cv2 BGR) -> dst float32 NHWC with ld = 4 (pad channel 0):
 * dst[c] = (src[c]/255 - mean3[c]) / std3[c].  mean3 / std3 are HOST pointers to 3 floats. */
int cvk_preprocess_u8(const uint8_t* src, float* dst, int N, int H, int W, const float* mean3, const float* std3, void* stream);

/* ---- fused AdamW over a flat fp32 buffer (torch.optim.AdamW: train.py:100,133) -------------------------------- */
int cvk_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);


/* ================================================================================================================
 * bf16-storage path (BASELINE.json configs[3] "bf16 + MFMA im2col path"; set_conv_precision(net, "bf16")).
 * Activations and activation gradients are bf16 NHWC in HBM, products bf16 x bf16 on the matrix cores with fp32
 * accumulation, BatchNorm statistics / parameter gradients / master weights fp32.  Same reference call sites as the
 * fp32 entry points above (models/unet.py:11-13,25,92; train.py:131).
 * ================================================================================================================ */

/* rows the packed weight tensors are padded to (multiple of the kernel's output-channel tile), and the number of
 * BatchNorm-statistics partials cvk_conv3x3_bf16s writes for a layer (one per 8 x 32 pixel tile, or one per strip of
 * tiles for the <= 64 x 64 channel layers); cvk_bf16s_stat_partials is the upper bound over all layers */
int cvk_bf16s_rows_pad(int cout);
int cvk_bf16s_stat_partials(int N, int H, int W);
int cvk_bf16s_stat_partials_c(int N, int H, int W, int Cin, int Cout);
/* fp32 master weights [Cout][3][3][Cin] -> bf16 [rows_pad(Cout)][9][Cin_pad] (forward), and the rotated/transposed
 * data-grad filter bf16 [rows_pad(Cin)][9][Cout_pad]; zero padded */
int cvk_pack_weight_fwd_bf16(const float* w, void* out, int Cout, int Cin, int Cin_pad, void* stream);
int cvk_pack_weight_dgrad_bf16(const float* w, void* out, int Cout, int Cin, int Cout_pad, void* stream);
/* n (<= CVK_PACK_BATCH_MAX) such packs in ONE launch: job i = cvk_pack_weight_fwd_bf16(w, out, Cout, Cin, Kpad) (dgrad = 0) or
 * cvk_pack_weight_dgrad_bf16(w, out, Cout, Cin, Kpad) (dgrad = 1); `jobs` is a HOST array (copied into the kernel arguments).
 * The executor packs every layer of a step up front when the weights have changed (models/unet.py:35-92: 23 conv layers). */
#define CVK_PACK_BATCH_MAX 48
typedef struct cvk_pack_job { const float* w; void* out; int Cout, Cin, Kpad, dgrad; } cvk_pack_job;
int cvk_pack_weights_bf16_batch(const cvk_pack_job* jobs, int n, void* stream);
/* y[N,H,W,ldy] (bf16) = conv3x3(x[N,H,W,Cin] bf16, w bf16 [rows_pad][9][Cin]) + bias; Cin % 32 == 0.  With stats != NULL:
 * stats[2][P][Cout] = per-tile (sum, M2 about the tile mean) of the fp32 results, counts[P] = pixels per tile,
 * P = cvk_bf16s_stat_partials_c(N,H,W,Cin,Cout) -> cvk_bn_finalize_counts.  Data-grad: the same call on dy and the dgrad pack. */
int cvk_conv3x3_bf16s(const void* x, const void* w, const float* bias, void* y, float* stats, float* counts,
                      int N, int H, int W, int Cin, int Cout, int ldy, void* stream);
/* the same with a cap on the workgroups of the persistent kernels (layers with >= 64 input and > 32 output channels run one
 * workgroup per CU that walks the tiles): under data-parallel training the executor leaves CVK_DP_RESERVE_CUS CUs to the RCCL
 * all-reduce kernels that run beside backward (legacy/train_tpu.py:115 hides the same exchange).  0 = every CU.  Results are
 * bitwise independent of the cap. */
int cvk_conv3x3_bf16s_wg(const void* x, const void* w, const float* bias, void* y, float* stats, float* counts,
                      int N, int H, int W, int Cin, int Cout, int ldy, int max_workgroups, void* stream);
/* The THIN layers of the bf16-storage mode on their own kernels (csrc/thin_bf16.hip: register-only, no LDS; round 5): the stem 3 -> 64
 * and the classifier head 64 -> class_num forward (models/unet.py:103,127) and the head's data-grad.  cvk_thin_bf16_mode(Cin, Cout of the
 * FORWARD layer, pitch of the kernel's input / output tensor, dgrad): 0 not thin, 1 head forward, 2 stem forward, 3 head data-grad.
 * cvk_pack_weight_thin_bf16: the forward layer's fp32 weights [Cout][3][3][Cin] -> the mode's filter pack (cvk_thin_bf16_pack_elems bf16).
 * cvk_conv3x3_thin_bf16: y bf16 [N,H,W,ld_out] = conv3x3(x bf16 [N,H,W,ld_in], pack) + bias; with stats != NULL statistics partials
 * stats[2][P][C] + counts[P], P = cvk_thin_bf16_stat_partials(N,H,W), C = Cout (mode 1) or 64 -> cvk_bn_finalize_counts. */
int cvk_thin_bf16_mode(int Cin, int Cout, int ld_in, int ld_out, int dgrad);
int cvk_thin_bf16_stat_partials(int N, int H, int W);
size_t cvk_thin_bf16_pack_elems(int mode);
int cvk_pack_weight_thin_bf16(const float* w, void* out, int Cout, int Cin, int mode, void* stream);
int cvk_conv3x3_thin_bf16(const void* x, const void* wpack, const float* bias, void* y, float* stats, float* counts, int N, int H, int W,
                          int ld_in, int Cout, int ld_out, int mode, void* stream);
/* which kernel the two calls above run for a layer geometry (a query of their dispatch, no launch; measurement tools label their
 * timings with the name a kernel trace shows): 0 k_conv_bf16s (tile kernel), 1 k_conv_bf16q, 2 k_conv_bf16h, 3 k_conv_bf16h on the
 * 128-row weight pack, 4 k_conv_bf16s_strip, 5 two k_conv_bf16s_strip passes (64 -> 128 channels without statistics); < 0: bad shape */
int cvk_conv3x3_bf16s_kernel(int N, int H, int W, int Cin, int Cout, int with_stats);
int cvk_bn_finalize_counts(const float* stats, const float* counts, int P, int M, int C, const float* gamma, const float* beta,
                           float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float momentum, float eps, void* workspace, size_t workspace_bytes,
                           void* stream);
/* dw fp32 [Cout][9][Cin] = sum_pixels dy (x) x_shifted; x bf16 [N,H,W,ldx], dy bf16 [N,H,W,ld_dy] (ld % 8 == 0) */
size_t cvk_conv3x3_wgrad_bf16s_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int cvk_conv3x3_wgrad_bf16s(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int ldx, int Cout,
                            int ld_dy, void* workspace, size_t workspace_bytes, void* stream);
/* The same in two steps, so that a backward pass can sum the partial results of ALL its layers in one launch at its end (the weight
 * gradients are read by nobody before the optimizer step / the all-reduce of their bucket: loss.backward() of train.py:131).
 * _slabs: the partial sums only — S = cvk_conv3x3_wgrad_bf16s_splits(...) slabs of Cout*9*Cin floats at `slabs`
 * (cvk_conv3x3_wgrad_bf16s_workspace_bytes of room); with S == 1 the one slab IS dw and may be written in place.
 * _reduce_batch: dw = slab 0 + slab 1 + ... (the order cvk_conv3x3_wgrad_bf16s uses: bitwise the same) for n <= CVK_WREDUCE_BATCH_MAX
 * layers; `jobs` is a HOST array (copied into the kernel arguments). */
int cvk_conv3x3_wgrad_bf16s_splits(int N, int H, int W, int Cin, int Cout);
int cvk_conv3x3_wgrad_bf16s_slabs(const void* x, const void* dy, float* slabs, int N, int H, int W, int Cin, int ldx, int Cout,
                                  int ld_dy, size_t slab_bytes, void* stream);
#define CVK_WREDUCE_BATCH_MAX 48
typedef struct cvk_wreduce_job { const float* slabs; float* dw; unsigned long long n; int splits, pad; } cvk_wreduce_job;
int cvk_wgrad_reduce_bf16s_batch(const cvk_wreduce_job* jobs, int n, void* stream);
/* logical NCHW fp32 (any strides) -> dense bf16 NHWC with ld % 8 == 0, pad channels zero */
int cvk_import_nchw_bf16(const float* src, int64_t sN, int64_t sC, int64_t sH, int64_t sW, void* dst, int ld,
                         int N, int C, int H, int W, void* stream);
/* out = relu(y*scale + shift): y bf16 [M][ldy]; out a view of bf16 (out_f32 = 0) or fp32 (out_f32 = 1, the logits);
 * pool != NULL: also writes MaxPool2d(2,2)(out) as dense bf16 [N,H/2,W/2,C] (models/unet.py:92 fused into this pass) */
int cvk_bn_relu_apply_bf16(const void* y, int ldy, const float* scale, const float* shift, cvk_viewh out, int out_f32,
                           void* pool, int N, int H, int W, int C, void* stream);
/* BatchNorm+ReLU backward on bf16 tensors; dout is a bf16 view, or fp32 (dout_f32 = 1: the loss gradient).
 * part: 2 * cvk_bn_bwd_blocks_bf16(M) * C floats (reduce) / cvk_bn_bwd_blocks_bf16(M) * C floats (dx: column sums of dy,
 * the conv bias gradient) -> cvk_colsum_finalize.  dy rows have pitch ld_dy >= C, pad columns are written as 0. */
int cvk_bn_bwd_blocks_bf16(int M);
int cvk_bn_bwd_reduce_bf16(cvk_viewh dout, int dout_f32, const void* y, int ldy, const float* scale, const float* shift,
                           const float* mean, const float* rstd, float* part, int N, int H, int W, int C, void* stream);
int cvk_bn_bwd_dx_bf16(cvk_viewh dout, int dout_f32, const void* y, int ldy, const float* scale, const float* shift,
                       const float* mean, const float* rstd, const float* dgamma, const float* dbeta, void* dy, int ld_dy,
                       float* dbias_part, int N, int H, int W, int C, int use_batch_stats, void* stream);
/* MaxPool2d(2,2) backward: dout dense bf16 [N,H/2,W/2,C]; x = the pooled layer's stored input (view); dx (view) is
 * written, or added to when accumulate != 0 */
int cvk_maxpool2x2_bwd_bf16(const void* dout, cvk_viewh x, cvk_viewh dx, int accumulate, int N, int H, int W, int C, void* stream);
/* cvk_maxpool2x2_bwd_bf16 that also leaves the partial sums of the producing block's BatchNorm+ReLU backward (reference models/unet.py:12-13 under
 * nn.MaxPool2d(2,2), unet.py:92; the bf16 twin of cvk_maxpool2x2_bwd_bnred): when the pool directly follows a conv block this pass is the last writer of
 * the block's output gradient, and it sums g = dx * [ReLU passed] and g * xhat over the STORED (bf16) values on the way — part = float[2]
 * [cvk_maxpool2x2_bwd_bnred_blocks_bf16(N,H,W,C)][C] for cvk_colsum_finalize; yP [N*H*W][ldp] bf16 = the block's conv output, scale / shift / mean /
 * rstd its BatchNorm constants (contract of cvk_bn_bwd_reduce_bf16).  C % 8 == 0 and C/8 must divide 256 (blocks() returns 0 otherwise). */
int cvk_maxpool2x2_bwd_bnred_blocks_bf16(int N, int H, int W, int C);
int cvk_maxpool2x2_bwd_bnred_bf16(const void* dout, cvk_viewh x, cvk_viewh dx, int accumulate, int N, int H, int W, int C, const void* yP, int ldp,
                                  const float* scale, const float* shift, const float* mean, const float* rstd, float* part, void* stream);
/* MaxUnpool2d(2) of bf16 plans (reference models/segnet.py:80,104-116: self.unpool(x, idx, output_size)).  No index tensor is kept:
 * the arg-max of every 2x2 cell is recomputed from x, the stored input of the pooling layer ([N,H,W,C] view; first maximum in scan
 * order, NaN wins — ATen's rule, as cvk_maxpool2x2_bwd_bf16).  FORWARD is cvk_maxpool2x2_bwd_bf16(v, x, out, 0, ...): the pooled
 * values v [N,H/2,W/2,C] scattered to their arg-max pixels, every other pixel 0.  BACKWARD (this entry) gathers:
 * dv[n,yc,xc,c] = dout[n, arg-max pixel of cell (yc,xc), c]; dout dense [N,H,W,C], dv dense [N,H/2,W/2,C].  C % 8 == 0. */
int cvk_maxunpool2x2_bwd_bf16(const void* dout, cvk_viewh x, void* dv, int N, int H, int W, int C, void* stream);
int cvk_bilinear_up2_fwd_bf16(const void* x, void* out, int N, int H, int W, int C, void* stream);
int cvk_bilinear_up2_bwd_bf16(const void* dout, void* dx, int N, int H, int W, int C, void* stream);
int cvk_zero_frame_bf16(cvk_viewh buf, int N, int H, int W, int C, int y0, int x0, int h, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVK_H */
