#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the UNet batch-8 shapes: fwd (with BN stats), dgrad, wgrad -> TFLOP/s."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

LAYERS = [  # name, Cin, Cout, H, W
    ("down1.0", 4, 64, 360, 480), ("down1.1", 64, 64, 360, 480), ("down2.0", 64, 128, 180, 240), ("down2.1", 128, 128, 180, 240),
    ("down3.0", 128, 256, 90, 120), ("down3.1", 256, 256, 90, 120), ("down4.0", 256, 512, 45, 60), ("down4.1", 512, 512, 45, 60),
    ("down5.0", 512, 1024, 22, 30), ("down5.1", 1024, 1024, 22, 30), ("ups1.conv", 1024, 512, 44, 60), ("up1.0", 1024, 512, 45, 60),
    ("ups2.conv", 512, 256, 90, 120), ("up2.0", 512, 256, 90, 120), ("ups3.conv", 256, 128, 180, 240), ("up3.0", 256, 128, 180, 240),
    ("ups4.conv", 128, 64, 360, 480), ("up4.0", 128, 64, 360, 480), ("output", 64, 12, 360, 480),
]

if "--rev" in sys.argv:        # data-grad shapes: channel roles exchanged
    sys.argv.remove("--rev")
    LAYERS = [(n, co, ci, h, w) for (n, ci, co, h, w) in LAYERS if ci >= 64 and co >= 64]
COLD = "--cold" in sys.argv
if COLD:
    sys.argv.remove("--cold")
_flush = None

def timeit(fn, n=5):
    global _flush
    fn(); torch.cuda.synchronize()
    if not COLD:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    if _flush is None:
        _flush = torch.empty(1 << 28, device="cuda")      # 1 GiB: evicts L2 (32 MiB) and the 256 MiB Infinity Cache
    tot = 0.0
    for _ in range(n):
        _flush.add_(1.0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e-3

def main():
    lib = _lib.load(); N = 8
    which = sys.argv[1:] or ["fwd", "wino", "dgrad", "wgrad", "wwino"]
    s = torch.cuda.current_stream().cuda_stream
    dev = "cuda"
    tot = {k: [0.0, 0.0] for k in which}
    for name, ci, co, H, W in LAYERS:
        M = N * H * W
        flops = 18.0 * M * ci * co
        x = torch.randn(M, ci, device=dev); w = torch.randn(co, 9 * ci, device=dev) * 0.05; b = torch.randn(co, device=dev)
        ldy = (co + 3) // 4 * 4
        y = torch.empty(M, ldy, device=dev); P = (M + 63) // 64
        stats = torch.empty(2 * P * co, device=dev)
        row = f"{name:10s} {ci:5d}->{co:5d} {H:3d}x{W:3d} "
        if "fwd" in which:
            t = timeit(lambda: check(lib.cvk_conv3x3_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, W, ci, co, ldy, 0, s)))
            row += f" fwd {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF"; tot["fwd"][0] += flops; tot["fwd"][1] += t
        if "wino" in which and ci % 64 == 0:
            U = torch.empty(4 * co * 3 * ci, device=dev)
            check(lib.cvk_wino_weight_transform(w.data_ptr(), U.data_ptr(), co, ci, s))
            wsb = lib.cvk_conv3x3_wino_workspace_bytes(N, H, W, ldy); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t = timeit(lambda: (check(lib.cvk_conv3x3_wino_gemm(x.data_ptr(), U.data_ptr(), ws.data_ptr(), N, H, W, ci, co, ldy, s)), check(lib.cvk_wino_output(ws.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, W, co, ldy, s))))
            row += f" wino {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF"; tot["wino"][0] += flops; tot["wino"][1] += t
        if "wino4" in which and ci % 64 == 0:
            yref = y.clone(); sref = stats.clone()
            U4 = torch.empty(6 * co * 3 * ci, device=dev)
            check(lib.cvk_wino4_weight_transform(w.data_ptr(), U4.data_ptr(), co, ci, s))
            wsb = lib.cvk_conv3x3_wino4_workspace_bytes(N, H, W, ci, ldy); ws4 = torch.empty(wsb, dtype=torch.uint8, device=dev)
            tg = timeit(lambda: check(lib.cvk_conv3x3_wino4_gemm(x.data_ptr(), U4.data_ptr(), ws4.data_ptr(), N, H, W, ci, co, ldy, 0, s)))
            to = timeit(lambda: check(lib.cvk_wino4_output(ws4.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), N, H, W, co, ldy, lib.cvk_conv3x3_wino4_ksplit(N, H, W, ci, ldy), s)))
            t = tg + to
            err = (y - yref).abs().max().item() / yref.abs().max().item() if "wino" in which else float("nan")
            serr = (stats - sref).abs().max().item() / sref.abs().max().item() if "wino" in which else float("nan")
            row += f" wino4 gemm {tg*1e6:7.1f} out {to*1e6:6.1f}us {flops/t/1e12:6.1f}TF (gemm {flops/tg/1e12:6.1f}) err {err:.1e} {serr:.1e}"; tot["wino4"][0] += flops; tot["wino4"][1] += t
        if "wino4f" in which and ci % 32 == 0 and co % 4 == 0 and ci <= 256 and co <= 256:
            yref = y.clone()
            Uf = torch.empty(lib.cvk_wino4f_weight_floats(co, ci), device=dev)
            check(lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), co, ci, 0, s))
            Pf = lib.cvk_wino4f_stat_partials(N, H, W); stf = torch.zeros(2 * Pf * co + Pf, device=dev)
            y.zero_()
            t = timeit(lambda: check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), stf.data_ptr(), stf.data_ptr() + 8 * Pf * co,
                                                            N, H, W, ci, co, ldy, 0, s)))
            err = (y - yref).abs().max().item() / max(yref.abs().max().item(), 1e-30) if "wino4" in which else float("nan")
            row += f" wino4f {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF err {err:.1e}"; tot["wino4f"][0] += flops; tot["wino4f"][1] += t
        if "w2d" in which and ci % 32 == 0 and co >= 64 and co % 4 == 0:
            yref = y.clone()
            U2 = torch.empty(36 * co * ci, device=dev)
            tw_ = timeit(lambda: check(lib.cvk_w2d_weight_transform(w.data_ptr(), U2.data_ptr(), co, ci, s)))
            wsb = lib.cvk_conv3x3_w2d_workspace_bytes(N, H, W, ci, co); ws2 = torch.empty(wsb, dtype=torch.uint8, device=dev)
            P2 = lib.cvk_w2d_stat_partials(N, H, W); st2 = torch.zeros(2 * P2 * co + P2, device=dev)
            y.zero_()
            t = timeit(lambda: check(lib.cvk_conv3x3_w2d(x.data_ptr(), U2.data_ptr(), b.data_ptr(), y.data_ptr(), st2.data_ptr(), st2.data_ptr() + 8 * P2 * co,
                                                         N, H, W, ci, co, ldy, ws2.data_ptr(), wsb, s)))
            err = (y - yref).abs().max().item() / max(yref.abs().max().item(), 1e-30)
            ssum = st2[:P2 * co].view(P2, co).sum(0); serr = (ssum - y[:, :co].sum(0)).abs().max().item() / y[:, :co].sum(0).abs().max().item()
            cnt = st2[2 * P2 * co:].sum().item()
            row += f" w2d {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF (wt {tw_*1e6:.1f}us) err {err:.1e} sum {serr:.1e} cnt {cnt:.0f}/{M}"; tot["w2d"][0] += flops; tot["w2d"][1] += t
        if "dgrad" in which and name != "down1.0":
            dy = torch.randn(M, ldy, device=dev); wd = torch.randn(ci, 9 * ldy, device=dev) * 0.05; dx = torch.empty(M, ci, device=dev)
            t = timeit(lambda: check(lib.cvk_conv3x3_fwd(dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, N, H, W, ldy, ci, ci, s)))
            row += f" dgrad {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF"; tot["dgrad"][0] += flops; tot["dgrad"][1] += t
        if "wwino" in which and ci >= 32 and co > 32:
            dy = torch.randn(M, ldy, device=dev); dw = torch.empty(co, 9 * ci, device=dev)
            wsb = lib.cvk_conv3x3_wgrad_wino_workspace_bytes(N, H, W, ci, co); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t = timeit(lambda: check(lib.cvk_conv3x3_wgrad_wino(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, ci, ci, co, ldy, ws.data_ptr(), wsb, s)))
            row += f" wwino {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF"; tot["wwino"][0] += flops; tot["wwino"][1] += t
        if "wwino4" in which and ci >= 32 and co > 32:
            dy = torch.randn(M, ldy, device=dev); dw4 = torch.empty(co, 9 * ci, device=dev)
            wsb = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, ci, co, ldy); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t = timeit(lambda: check(lib.cvk_conv3x3_wgrad_wino4(x.data_ptr(), dy.data_ptr(), None, dw4.data_ptr(), N, H, W, ci, ci, co, ldy, ws.data_ptr(), wsb, s)))
            dwr = torch.empty(co, 9 * ci, device=dev)
            wsb2 = lib.cvk_conv3x3_wgrad_wino_workspace_bytes(N, H, W, ci, co); ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
            check(lib.cvk_conv3x3_wgrad_wino(x.data_ptr(), dy.data_ptr(), dwr.data_ptr(), N, H, W, ci, ci, co, ldy, ws2.data_ptr(), wsb2, s))
            err = (dw4 - dwr).abs().max().item() / dwr.abs().max().item()
            row += f" wwino4 {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF err {err:.1e}"; tot["wwino4"][0] += flops; tot["wwino4"][1] += t
        if "wgradp" in which and ci % 64 == 0 and co % 64 == 0 and ci <= 256 and co <= 256:
            dy = torch.randn(M, ldy, device=dev); dwf = torch.empty(co, 9 * ci, device=dev)
            wsb = lib.cvk_conv3x3_wgradp_workspace_bytes(N, H, W, ci, co); wsf = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t = timeit(lambda: check(lib.cvk_conv3x3_wgradp(x.data_ptr(), dy.data_ptr(), None, dwf.data_ptr(), N, H, W, ci, ci, co, ldy, wsf.data_ptr(), wsb, s)))
            dwr = torch.empty(co, 9 * ci, device=dev)
            wsb2 = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, ci, co, ldy); ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
            t4 = timeit(lambda: check(lib.cvk_conv3x3_wgrad_wino4(x.data_ptr(), dy.data_ptr(), None, dwr.data_ptr(), N, H, W, ci, ci, co, ldy, ws2.data_ptr(), wsb2, s)))
            err = (dwf - dwr).norm().item() / dwr.norm().item()
            row += f" wgradp {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF (wwino4 {t4*1e6:8.1f}us, ratio {t/t4:.2f}) relL2 {err:.1e}"; tot["wgradp"][0] += flops; tot["wgradp"][1] += t
        if "ww2d" in which and ci % 4 == 0 and co % 4 == 0 and ci >= 32 and co > 32:
            dy = torch.randn(M, ldy, device=dev); dw2 = torch.empty(co, 9 * ci, device=dev)
            wsb = lib.cvk_conv3x3_wgrad_w2d_workspace_bytes(N, H, W, ci, co); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t = timeit(lambda: check(lib.cvk_conv3x3_wgrad_w2d(x.data_ptr(), dy.data_ptr(), dw2.data_ptr(), N, H, W, ci, ci, co, ldy, ws.data_ptr(), wsb, s)))
            dwr = torch.empty(co, 9 * ci, device=dev)
            wsb2 = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, ci, co, ldy); ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
            t4 = timeit(lambda: check(lib.cvk_conv3x3_wgrad_wino4(x.data_ptr(), dy.data_ptr(), None, dwr.data_ptr(), N, H, W, ci, ci, co, ldy, ws2.data_ptr(), wsb2, s)))
            err = (dw2 - dwr).norm().item() / dwr.norm().item()
            row += f" ww2d {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF (wwino4 {t4*1e6:8.1f}us, ratio {t/t4:.2f}) relL2 {err:.1e}"; tot["ww2d"][0] += flops; tot["ww2d"][1] += t
        if "wgrad" in which:
            dy = torch.randn(M, ldy, device=dev); dw = torch.empty(co, 9 * ci, device=dev)
            wsb = lib.cvk_conv3x3_wgrad_workspace_bytes(N, H, W, ci, co); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t = timeit(lambda: check(lib.cvk_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, ci, ci, co, ldy, ws.data_ptr(), wsb, s)))
            row += f" wgrad {t*1e6:8.1f}us {flops/t/1e12:6.1f}TF"; tot["wgrad"][0] += flops; tot["wgrad"][1] += t
        print(row, flush=True)
    for k, (f, t) in tot.items():
        print(f"TOTAL {k}: {t*1e3:.2f} ms  {f/t/1e12:.1f} TF")

main()
