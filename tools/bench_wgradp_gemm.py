#!/usr/bin/env python3
"""cvk_wgradp_gemm alone (E6 / V6 planes given) against cvk_conv3x3_wgrad_wino4 (E planes given, V transformed in the kernel) at the layers
that still run the latter — what free V planes (written by the forward kernel) would buy.  Cold: L2 / Infinity Cache flushed between runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

LAYERS = [("ups4.conv", 128, 64, 360, 480), ("up4.0", 128, 64, 360, 480), ("down2.1", 128, 128, 180, 240), ("down3.0", 128, 256, 90, 120),
          ("down1.1", 64, 64, 360, 480), ("down2.0", 64, 128, 180, 240)]
_flush = None


def cold(fn, n=5):
    global _flush
    if _flush is None:
        _flush = torch.empty(1 << 28, device="cuda")
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        _flush.add_(1.0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


def main():
    lib = _lib.load(); N = 8
    s = torch.cuda.current_stream().cuda_stream
    for name, ci, co, H, W in LAYERS:
        M = N * H * W
        x = torch.randn(M, ci, device="cuda"); dy = torch.randn(M, co, device="cuda")
        rows = lib.cvk_wgradp_plane_rows(N, H, W)
        E6 = torch.empty(6 * rows * co, device="cuda"); V6 = torch.empty(6 * rows * ci, device="cuda")
        check(lib.cvk_wgradp_planes(x.data_ptr(), ci, V6.data_ptr(), N, H, W, ci, 0, s))
        check(lib.cvk_wgradp_planes(dy.data_ptr(), co, E6.data_ptr(), N, H, W, co, 1, s))
        wsb = lib.cvk_wgradp_gemm_workspace_bytes(N, H, W, ci, co); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        dw = torch.empty(co, 9 * ci, device="cuda")
        tg = cold(lambda: check(lib.cvk_wgradp_gemm(E6.data_ptr(), V6.data_ptr(), dw.data_ptr(), N, H, W, ci, ci, co, ws.data_ptr(), wsb, s)))
        tp = cold(lambda: check(lib.cvk_wgradp_planes(x.data_ptr(), ci, V6.data_ptr(), N, H, W, ci, 0, s)))
        wsb2 = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, ci, co, co); ws2 = torch.empty(wsb2, dtype=torch.uint8, device="cuda")
        Wt = (W + 3) // 4
        E4 = torch.randn(4 * N * H * Wt * co, device="cuda")
        dw2 = torch.empty(co, 9 * ci, device="cuda")
        t4 = cold(lambda: check(lib.cvk_conv3x3_wgrad_wino4(x.data_ptr(), dy.data_ptr(), E4.data_ptr(), dw2.data_ptr(), N, H, W, ci, ci, co, co, ws2.data_ptr(), wsb2, s)))
        fl = 9.0 * M * ci * co
        print(f"{name:10s} {ci:4d}->{co:4d} {H}x{W}: wgradp gemm {tg:7.1f} us ({fl / tg / 1e6 / 157.3:.2f} executed)  V planes {tp:6.1f} us   wgrad_wino4 {t4:7.1f} us ({fl / t4 / 1e6 / 157.3:.2f})", flush=True)


if __name__ == "__main__":
    main()
