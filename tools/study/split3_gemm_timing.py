#!/usr/bin/env python3
"""time ONE layer's GEMM stage with a real micro-kernel — cvk_w2d_gemm_split3 (csrc/split3.hip: 3-term split fp32 operands
on the bf16 matrix pipe) against the exact-fp32 GEMM the product runs (cvk_w6_gemm), same shapes, F(6x6,3x3) planes (64 GEMMs), plus the
cost of the stand-alone conversion pass (cvk_split3_planes; a production version would emit the terms from the transform kernels).
    python tools/study/split3_gemm_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_camvid_amd import _lib          # noqa: E402
from pytorch_camvid_amd._lib import check    # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
N, NX = 8, 64
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


print("layer (batch 8, F(6x6,3x3))      fp32 GEMM    split-3 GEMM   speed-up   (V conversion pass)   executed bf16 TFLOP/s")
t32s = t3s = 0.0
for name, ci, co, h, w in LAYERS:
    T = lib.cvk_w6_tiles(N, h, w)
    Tp32 = lib.cvk_w2d_tpad(T)
    V = torch.randn(NX * Tp32 * ci + 128, device=dev)
    U = torch.randn(NX * co * ci, device=dev) * 0.05
    f = lib.cvk_w6_ksplit(T, ci, co)
    Mo = torch.empty(f * NX * T * co + 1024, device=dev)
    t32 = timeit(lambda: check(lib.cvk_w6_gemm(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), T, ci, co, s)))
    Tp, Cp = lib.cvk_split3_rows_pad(T, 256), lib.cvk_split3_rows_pad(co, 128)
    Vv = V[:NX * Tp32 * ci].view(NX, Tp32, ci)[:, :T].contiguous()
    V3 = torch.empty(NX * (ci // 32) * 3 * Tp * 32, device=dev, dtype=torch.bfloat16)
    U3 = torch.empty(NX * (ci // 32) * 3 * Cp * 32, device=dev, dtype=torch.bfloat16)
    tc = timeit(lambda: check(lib.cvk_split3_planes(Vv.data_ptr(), V3.data_ptr(), NX, T, Tp, ci, s)))
    check(lib.cvk_split3_planes(U.data_ptr(), U3.data_ptr(), NX, co, Cp, ci, s))
    t3 = timeit(lambda: check(lib.cvk_w2d_gemm_split3(V3.data_ptr(), U3.data_ptr(), Mo.data_ptr(), NX, T, Tp, ci, co, Cp, s)))
    t32s += t32; t3s += t3
    print(f"{name:8s} {ci:5d}->{co:4d} @{h:3d}x{w:3d}   {t32*1e6:8.1f} us   {t3*1e6:8.1f} us     {t32/t3:5.2f}x      {tc*1e6:8.1f} us          {12.0*NX*Tp*ci*Cp/t3/1e12:8.0f}")
    del V, U, Mo, V3, U3, Vv
print(f"sum                            {t32s*1e6:8.1f} us   {t3s*1e6:8.1f} us     {t32s/t3s:5.2f}x")

# the second format (two scaled fp16 terms, three cross-products): the same GEMMs; "streamed" = operand bytes the workgroups pull through
# their LDS (V once per Cout tile, U once per row tile) + the product planes written, over the launch time
print()
print("layer (batch 8, F(6x6,3x3))      split-3 GEMM   split-2h GEMM   executed fp16 TFLOP/s   streamed GB/s (fmt 3 / fmt 2)")
for name, ci, co, h, w in LAYERS:
    T = lib.cvk_w6_tiles(N, h, w)
    Tp, Cp = lib.cvk_split3_rows_pad(T, 256), lib.cvk_split3_rows_pad(co, 128)
    Mo = torch.empty(NX * T * co + 1024, device=dev)
    am = torch.full((lib.cvk_amax_block_words(),), 0x3F800000, device=dev, dtype=torch.int32)
    res = {}
    for fmt in (3, 2):
        dt = torch.bfloat16 if fmt == 3 else torch.float16
        Vs = (torch.randn(NX * (ci // 32) * fmt * Tp * 32, device=dev) * 0.5).to(dt)
        Us = (torch.randn(NX * (ci // 32) * fmt * Cp * 32, device=dev) * 0.05).to(dt)
        t = timeit(lambda: check(lib.cvk_w2d_gemm_split(fmt, 6, Vs.data_ptr(), Us.data_ptr(), Mo.data_ptr(), am.data_ptr(), am.data_ptr(), NX, T, Tp, ci, co, Cp, s)))
        streamed = NX * (2.0 * fmt * ci * (Tp * (Cp // 128) + Cp * (Tp // 256)) + 4.0 * T * co)
        res[fmt] = (t, streamed / t / 1e9)
        del Vs, Us
    print(f"{name:8s} {ci:5d}->{co:4d} @{h:3d}x{w:3d}   {res[3][0]*1e6:8.1f} us   {res[2][0]*1e6:8.1f} us     {6.0*NX*Tp*ci*Cp/res[2][0]/1e12:8.0f}               "
          f"{res[3][1]:6.0f} / {res[2][1]:6.0f}")
    del Mo

print()
print("weight-grad GEMM (k = tiles)        fp32 gemm_tn   split-3 gemm_tn   speed-up")
t32s = t3s = 0.0
for name, ci, co, h, w in LAYERS:
    T = lib.cvk_w6_tiles(N, h, w)
    Tp32 = lib.cvk_w2d_tpad(T); Tp = lib.cvk_split3_rows_pad(T, 256)
    E = torch.randn(NX * Tp32 * co + 128, device=dev) * 0.1
    V = torch.randn(NX * Tp32 * ci + 128, device=dev)
    f32 = lib.cvk_w6_wgrad_ksplit(T, ci, co)
    P = torch.empty(max(f32, 16) * NX * co * ci, device=dev)
    t32 = timeit(lambda: check(lib.cvk_w6_gemm_tn(E.data_ptr(), V.data_ptr(), P.data_ptr(), T, ci, co, s)))
    E3 = torch.empty(NX * (co // 32) * 3 * Tp * 32, device=dev, dtype=torch.bfloat16)
    V3 = torch.empty(NX * (ci // 32) * 3 * Tp * 32, device=dev, dtype=torch.bfloat16)
    check(lib.cvk_split3_planes(E[:NX * Tp32 * co].view(NX, Tp32, co)[:, :T].contiguous().data_ptr(), E3.data_ptr(), NX, T, Tp, co, s))
    check(lib.cvk_split3_planes(V[:NX * Tp32 * ci].view(NX, Tp32, ci)[:, :T].contiguous().data_ptr(), V3.data_ptr(), NX, T, Tp, ci, s))
    t3 = timeit(lambda: check(lib.cvk_w2d_gemm_tn_split3(E3.data_ptr(), V3.data_ptr(), P.data_ptr(), NX, Tp, ci, co, s)))
    t32s += t32; t3s += t3
    print(f"{name:8s} {ci:5d}->{co:4d} @{h:3d}x{w:3d}   {t32*1e6:8.1f} us   {t3*1e6:8.1f} us     {t32/t3:5.2f}x")
    del E, V, P, E3, V3
print(f"sum                            {t32s*1e6:8.1f} us   {t3s*1e6:8.1f} us     {t32s/t3s:5.2f}x")
