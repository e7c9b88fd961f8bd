#!/usr/bin/env python3
"""VERDICT r4 #8 (study only): what would the GEMM stage of the 2-D Winograd path cost on the bf16 matrix pipe with 3-term split
operands (6 bf16 products per fp32 product, tools/study/split_bf16_model.py)?  No such kernel exists; the MEASURED proxy is the bf16
implicit-GEMM convolution kernel of configs[3] (k_conv_bf16q / k_conv_bf16h: LDS-DMA staged bf16 operands, v_mfma_f32_16x16x32_bf16,
fp32 accumulators) at the same channel counts: a split F(6x6,3x3) GEMM stage issues 64 T Cin Cout x 6 multiply-adds, the 3x3
convolution of P pixels issues 9 P Cin Cout, so  t_split ~ t_conv_bf16 x (64 x 6 T) / (9 P)  at equal kernel efficiency.
Compared with the time of the exact-fp32 batched GEMM the product runs today (cvk_w6_gemm, k_w2d_gemm<128,32,2,2>).
    python tools/study/split_bf16_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_camvid_amd import _lib          # noqa: E402
from pytorch_camvid_amd._lib import check    # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
N = 8
LAYERS = [("down3.1", 256, 256, 90, 120), ("up2.0", 512, 256, 90, 120), ("down4.1", 512, 512, 45, 60), ("up1.0", 1024, 512, 45, 60),
          ("up3.0", 256, 128, 180, 240)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


print("layer (batch 8)            fp32 F(6x6) GEMM stage   bf16 conv proxy   split-F(6x6) GEMM estimate   ratio fp32 / split")
tot32 = tots = 0.0
for name, ci, co, h, w in LAYERS:
    T = lib.cvk_w6_tiles(N, h, w)
    Tp = lib.cvk_w2d_tpad(T)
    V = torch.randn(64 * Tp * ci + 128, device=dev)
    U = torch.randn(64 * co * ci, device=dev) * 0.05
    Mo = torch.empty(4 * 64 * Tp * co + 1024, device=dev)          # room for the K-split partial planes
    t32 = timeit(lambda: check(lib.cvk_w6_gemm(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), T, ci, co, s)))
    x = torch.randn(N, h, w, ci, device=dev).to(torch.bfloat16)
    wt = torch.randn(co, 3, 3, ci, device=dev) * 0.05
    wp = torch.empty(lib.cvk_bf16s_rows_pad(co) * 9 * ci, device=dev, dtype=torch.bfloat16)
    check(lib.cvk_pack_weight_fwd_bf16(wt.data_ptr(), wp.data_ptr(), co, ci, ci, s))
    y = torch.empty(N * h * w * co, device=dev, dtype=torch.bfloat16)
    tb = timeit(lambda: check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, None, N, h, w, ci, co, co, s)))
    P = N * h * w
    est = tb * (64.0 * 6.0 * T) / (9.0 * P)
    tot32 += t32; tots += est
    print(f"{name:8s} {ci:5d}->{co:4d} @{h:3d}x{w:3d}   {t32*1e6:8.1f} us            {tb*1e6:8.1f} us        {est*1e6:8.1f} us                  {t32/est:5.2f}x")
    del V, U, Mo, x, y
print(f"sum                          {tot32*1e6:8.1f} us                                   {tots*1e6:8.1f} us                  {tot32/tots:5.2f}x")
