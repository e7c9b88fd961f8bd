#!/usr/bin/env python3
"""The batched GEMM of the 2-D Winograd path alone (cvk_w6_gemm = k_w2d_gemm, F(6x6,3x3): 64 transform planes), per layer of the headline step
(UNet 8x3x360x480: the 13 layers the 2-D path takes, forward and data-grad): microseconds and the executed fraction of the fp32 matrix
peak.  With the experiments library (CVK_LIB_PATH=.../libcvk_exp.so) CVK_W2D_ABL=1..5 selects an ablation variant (WRONG results):
1 no barrier, 2 no DMA in the K loop, 3 no stores, 4 no LDS reads, 5 MFMA only.        usage (GPU box): python tools/bench_w2d_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

PEAK = 157.3e12
NX = 64
LAYERS = [  # name, Cin, Cout, H, W  (forward orientation)
    ("down3.0", 128, 256, 90, 120), ("down3.1", 256, 256, 90, 120), ("down4.0", 256, 512, 45, 60), ("down4.1", 512, 512, 45, 60),
    ("down5.0", 512, 1024, 22, 30), ("down5.1", 1024, 1024, 22, 30), ("ups1.conv", 1024, 512, 44, 60), ("up1.0", 1024, 512, 45, 60),
    ("ups2.conv", 512, 256, 90, 120), ("up2.0", 512, 256, 90, 120), ("ups3.conv", 256, 128, 180, 240), ("up3.0", 256, 128, 180, 240),
    ("down2.1*", 128, 128, 180, 240),
]


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3        # us


def main():
    lib = _lib.load()
    N = 8
    s = torch.cuda.current_stream().cuda_stream
    tot_t = tot_f = 0.0
    for mode in ("fwd", "rev"):
        for name, ci, co, H, W in LAYERS:
            if mode == "rev":
                ci, co = co, ci
            T = lib.cvk_w6_tiles(N, H, W); Tp = lib.cvk_w2d_tpad(T); ks = lib.cvk_w6_ksplit(T, ci, co)
            V = torch.randn(NX * Tp * ci + 128, device="cuda"); U = torch.randn(NX * co * ci, device="cuda") * 0.05
            Mo = torch.empty(ks * NX * T * co, device="cuda")
            t = timeit(lambda: check(lib.cvk_w6_gemm(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), T, ci, co, s)))
            fl = 2.0 * NX * T * ci * co
            if not name.endswith("*"):
                tot_t += t; tot_f += fl
            print(f"{mode} {name:10s} {ci:5d}->{co:5d} {H:3d}x{W:3d} T={T:6d} ksplit={ks}  {t:7.1f} us  {fl / t / 1e6 / PEAK * 1e12:.3f}", flush=True)
    print(f"sum (without *) {tot_t / 1e3:.3f} ms  {tot_f / tot_t / 1e6 / PEAK * 1e12:.4f} of {PEAK / 1e12:.1f} TFLOP/s")
    tot_t = tot_f = 0.0
    for name, ci, co, H, W in LAYERS:          # the transposed GEMM of the weight-grad (k_w2d_gemm_tn): depth = tiles
        T = lib.cvk_w6_tiles(N, H, W); Tp = lib.cvk_w2d_tpad(T); fs = lib.cvk_w6_wgrad_ksplit(T, ci, co)
        V = torch.randn(NX * Tp * ci + 128, device="cuda"); E = torch.randn(NX * Tp * co + 128, device="cuda")
        P = torch.empty(fs * NX * co * ci, device="cuda")
        t = timeit(lambda: check(lib.cvk_w6_gemm_tn(E.data_ptr(), V.data_ptr(), P.data_ptr(), T, ci, co, s)))
        fl = 2.0 * NX * T * ci * co
        if not name.endswith("*"):
            tot_t += t; tot_f += fl
        print(f"tn  {name:10s} {ci:5d}->{co:5d} {H:3d}x{W:3d} T={T:6d} ksplit={fs}  {t:7.1f} us  {fl / t / 1e6 / PEAK * 1e12:.3f}", flush=True)
    print(f"sum tn (without *) {tot_t / 1e3:.3f} ms  {tot_f / tot_t / 1e6 / PEAK * 1e12:.4f} of {PEAK / 1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
