#!/usr/bin/env python3
"""Per-layer timing at the UNet batch-8 shapes: fused 1-D F(4,3) (cvk_conv3x3_wino4f) vs the 2-D pipelines F(4x4,3x3) (cvk_w2d_*)
and F(6x6,3x3) (cvk_w6_*) for forward / data-grad, and transposed F(4,3) (cvk_conv3x3_wgrad_wino4) vs the 2-D weight-grads.
Feeds engine.py wino2d_pays / wgrad2d_pays.       usage (GPU box): python tools/bench_w6.py [fwd] [rev] [wgrad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

LAYERS = [  # name, Cin, Cout, H, W
    ("down1.1", 64, 64, 360, 480), ("down2.0", 64, 128, 180, 240), ("down2.1", 128, 128, 180, 240),
    ("down3.0", 128, 256, 90, 120), ("down3.1", 256, 256, 90, 120), ("down4.0", 256, 512, 45, 60), ("down4.1", 512, 512, 45, 60),
    ("down5.0", 512, 1024, 22, 30), ("down5.1", 1024, 1024, 22, 30), ("ups1.conv", 1024, 512, 44, 60), ("up1.0", 1024, 512, 45, 60),
    ("ups2.conv", 512, 256, 90, 120), ("up2.0", 512, 256, 90, 120), ("ups3.conv", 256, 128, 180, 240), ("up3.0", 256, 128, 180, 240),
    ("ups4.conv", 128, 64, 360, 480), ("up4.0", 128, 64, 360, 480),
]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3        # us


def main():
    lib = _lib.load()
    N = 8
    which = sys.argv[1:] or ["fwd", "rev", "wgrad"]
    s = torch.cuda.current_stream().cuda_stream
    for mode in which:
        print(f"== {mode}")
        for name, ci, co, H, W in LAYERS:
            if mode == "rev":
                ci, co = co, ci
            M = N * H * W
            x = torch.randn(M, ci, device="cuda")
            w = torch.randn(co, 9 * ci, device="cuda") * 0.05
            b = torch.randn(co, device="cuda")
            row = f"{name:10s} {ci:5d}->{co:5d} {H:3d}x{W:3d} "
            if mode in ("fwd", "rev"):
                y = torch.empty(M, co, device="cuda")
                if ci % 32 == 0 and co >= 32 and ci <= 256 and co <= 256:
                    Uf = torch.empty(lib.cvk_wino4f_weight_floats(co, ci), device="cuda")
                    check(lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), co, ci, 0, s))
                    Pf = lib.cvk_wino4f_stat_partials(N, H, W)
                    st = torch.zeros(2 * Pf * co + Pf, device="cuda")
                    sp = st.data_ptr() if mode == "fwd" else None
                    t = timeit(lambda: check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), sp,
                                                                   st.data_ptr() + 8 * Pf * co if sp else None, N, H, W, ci, co, co, 0, s)))
                    row += f" wino4f {t:8.1f}"
                else:
                    row += " wino4f        -"
                for mt, pre in ((4, "cvk_w2d_"), (6, "cvk_w6_")):
                    f = lambda n_: getattr(lib, pre + n_)
                    nx = (mt + 2) ** 2
                    U = torch.empty(nx * co * ci, device="cuda")
                    tw = timeit(lambda: check(f("weight_transform")(w.data_ptr(), U.data_ptr(), co, ci, s)))
                    T = f("tiles")(N, H, W); Tp = lib.cvk_w2d_tpad(T); ks = f("ksplit")(T, ci, co)
                    V = torch.empty(nx * Tp * ci + 128, device="cuda"); Mo = torch.empty(ks * nx * T * co, device="cuda")
                    P = f("stat_partials")(N, H, W); st = torch.zeros(2 * P * co + P, device="cuda")
                    sp = st.data_ptr() if mode == "fwd" else None
                    ti = timeit(lambda: check(f("input_transform")(x.data_ptr(), V.data_ptr(), N, H, W, ci, s)))
                    tg = timeit(lambda: check(f("gemm")(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), T, ci, co, s)))
                    to = timeit(lambda: check(f("output")(Mo.data_ptr(), b.data_ptr(), y.data_ptr(), sp, st.data_ptr() + 8 * P * co if sp else None,
                                                          N, H, W, ci, co, co, s)))
                    row += f" | F{mt}: in {ti:6.1f} gemm {tg:7.1f} out {to:6.1f} = {ti + tg + to:7.1f} (+wt {tw:5.1f})"
            else:
                dy = torch.randn(M, co, device="cuda")
                dw = torch.empty(co, 9 * ci, device="cuda")
                wsb = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, ci, co, co)
                ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
                t4 = timeit(lambda: check(lib.cvk_conv3x3_wgrad_wino4(x.data_ptr(), dy.data_ptr(), None, dw.data_ptr(), N, H, W, ci, ci, co, co,
                                                                     ws.data_ptr(), wsb, s)))
                row += f" wgrad_wino4 {t4:8.1f}"
                for mt, pre in ((4, "cvk_w2d_"), (6, "cvk_w6_")):
                    f = lambda n_: getattr(lib, pre + n_)
                    nx = (mt + 2) ** 2
                    T = f("tiles")(N, H, W); Tp = lib.cvk_w2d_tpad(T); fs = f("wgrad_ksplit")(T, ci, co)
                    V = torch.empty(nx * Tp * ci + 128, device="cuda"); E = torch.empty(nx * Tp * co + 128, device="cuda")
                    Pp = torch.empty(fs * nx * co * ci, device="cuda")
                    ti = timeit(lambda: check(f("input_transform")(x.data_ptr(), V.data_ptr(), N, H, W, ci, s)))
                    td = timeit(lambda: check(f("dy_transform")(dy.data_ptr(), co, E.data_ptr(), N, H, W, co, s)))
                    tg = timeit(lambda: check(f("gemm_tn")(E.data_ptr(), V.data_ptr(), Pp.data_ptr(), T, ci, co, s)))
                    to = timeit(lambda: check(f("wgrad_output")(Pp.data_ptr(), dw.data_ptr(), T, ci, ci, co, s)))
                    row += f" | F{mt}: x {ti:6.1f} dy {td:6.1f} gemm {tg:7.1f} out {to:5.1f} = {td + tg + to:7.1f} (+x {ti + td + tg + to:7.1f})"
            print(row, flush=True)


if __name__ == "__main__":
    main()
