// Internal interface between conv_bf16s.hip (C-ABI entry points, layer dispatch) and conv_bf16p.hip (the ping-pong kernel).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cvk.h"

namespace cvk_bf16p {

constexpr int TH = 16, TW = 32;     // output tile of one workgroup: 16 rows x 32 columns = 512 pixels
constexpr int BN = 128;             // output channels of one workgroup
constexpr int CK = 32;              // input channels per K slice

// Layers the ping-pong kernel serves (the weight pack for them is tile-major, see k_pack_w_pp): more than 64 output channels
// (a full 128-row weight tile) and at least 128 input channels (>= 4 slices: the K loop amortises prologue and epilogue).
// A function of the channel counts only — the pack functions have no geometry.
bool serves(int Cin, int Cout);
int kind(int Cin, int Cout);        // 0: not served, 1: 128 output channels per workgroup, 2: 64
// the kernel launch() runs for a layer geometry: 0 not served, 1 k_conv_bf16q, 2 k_conv_bf16h, 3 k_conv_bf16h on the 128-row pack
int choose(int N, int H, int W, int Cin, int Cout);

int stat_partials(int N, int H, int W);

// y = conv3x3(x, wpp) (+ bias) (+ statistics partials): same contract as cvk_conv3x3_bf16s, weights in the tile-major pack
// max_workgroups > 0 caps the grid of the persistent kernels (data parallel: CUs left to the collectives); results do not depend on it
void launch(const void* x, const void* wpp, const float* bias, void* y, float* stats, float* counts, int N, int H, int W, int Cin,
            int Cout, int ldy, hipStream_t s, int max_workgroups = 0);

// fp32 master [Cout][3][3][Cin] -> tile-major bf16 pack; dgrad = rotated by 180 degrees, channels exchanged
void pack(const float* w, void* out, int Cout, int Cin, int Kpad, bool dgrad, hipStream_t s);

// every pack of a step in one launch (n <= CVK_PACK_BATCH_MAX); layouts as pack() / cvk_pack_weight_{fwd,dgrad}_bf16 choose them
void pack_batch(const cvk_pack_job* jobs, int n, hipStream_t s);

}  // namespace cvk_bf16p
