#!/usr/bin/env python3
"""Round 6: what the forward-emitted V planes cost the fused forward launch and what they buy the weight-grad, per layer of the headline step
(cold caches between runs, like tools/bench_wgradp_gemm.py).  Forward: cvk_conv3x3_wino4f vs cvk_conv3x3_wino4f_vplanes.  Weight-grad: the plane GEMM
on forward-emitted slice-major planes (cvk_wgradp_gemm_sm) vs today's path of that layer (64 input channels: cvk_wgradp_planes + cvk_wgradp_gemm;
128: cvk_conv3x3_wgrad_wino4 with E planes given)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

LAYERS = [("down1.1", 64, 64, 360, 480), ("up4.1", 64, 64, 360, 480), ("ups4.conv", 128, 64, 360, 480), ("up4.0", 128, 64, 360, 480),
          ("down2.0", 64, 128, 180, 240), ("down2.1", 128, 128, 180, 240), ("up3.1", 128, 128, 180, 240), ("down3.0", 128, 256, 90, 120)]
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
    base = None         # optional: an older build of the library (raw handle), e.g. scratch/baselib/libcvk_r6base.so = the kernels before round 6
    if len(sys.argv) > 1:
        import ctypes
        base = ctypes.CDLL(os.path.abspath(sys.argv[1]))
        base.cvk_conv3x3_wino4f.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    tf = tv = tw_old = tw_new = 0.0
    for name, ci, co, H, W in LAYERS:
        M = N * H * W
        x = torch.randn(M, ci, device="cuda"); dy = torch.randn(M, co, device="cuda")
        w = torch.randn(co, 9 * ci, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
        y = torch.empty(M, co, device="cuda")
        Uf = torch.empty(lib.cvk_wino4f_weight_floats(co, ci), device="cuda")
        check(lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), co, ci, 0, s))
        P = lib.cvk_wino4f_stat_partials(N, H, W); st = torch.zeros(2 * P * co + P, device="cuda")
        rows = lib.cvk_wgradp_plane_rows(N, H, W)
        V6 = torch.empty(6 * rows * ci, device="cuda"); E6 = torch.empty(6 * rows * co, device="cuda")
        check(lib.cvk_wgradp_zero_pads_sm(V6.data_ptr(), N, H, W, ci, s))
        check(lib.cvk_wgradp_planes(dy.data_ptr(), co, E6.data_ptr(), N, H, W, co, 1, s))
        t_plain = cold(lambda: check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co,
                                                            N, H, W, ci, co, co, 0, s)))
        t_vpl = cold(lambda: check(lib.cvk_conv3x3_wino4f_vplanes(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co,
                                                                  V6.data_ptr(), N, H, W, ci, co, co, 0, s)))
        t_base = cold(lambda: base.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co,
                                                      N, H, W, ci, co, co, 0, s)) if base is not None else float("nan")
        wsb = lib.cvk_wgradp_gemm_workspace_bytes(N, H, W, ci, co); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        dw = torch.empty(co, 9 * ci, device="cuda")
        t_new = cold(lambda: check(lib.cvk_wgradp_gemm_sm(E6.data_ptr(), V6.data_ptr(), dw.data_ptr(), N, H, W, ci, ci, co, ws.data_ptr(), wsb, s)))
        if ci == 64:
            Vr = torch.empty(6 * rows * ci, device="cuda")
            def old():
                check(lib.cvk_wgradp_planes(x.data_ptr(), ci, Vr.data_ptr(), N, H, W, ci, 0, s))
                check(lib.cvk_wgradp_gemm(E6.data_ptr(), Vr.data_ptr(), dw.data_ptr(), N, H, W, ci, ci, co, ws.data_ptr(), wsb, s))
            t_old = cold(old)
        else:
            wsb2 = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, ci, co, co); ws2 = torch.empty(wsb2, dtype=torch.uint8, device="cuda")
            E4 = torch.randn(4 * N * H * ((W + 3) // 4) * co, device="cuda")
            t_old = cold(lambda: check(lib.cvk_conv3x3_wgrad_wino4(x.data_ptr(), dy.data_ptr(), E4.data_ptr(), dw.data_ptr(), N, H, W, ci, ci, co, co,
                                                                   ws2.data_ptr(), wsb2, s)))
        fl = 9.0 * M * ci * co
        tf += t_plain; tv += t_vpl; tw_old += t_old; tw_new += t_new
        print(f"{name:10s} {ci:4d}->{co:4d} {H}x{W}: forward (older build {t_base:7.1f} us) {t_plain:7.1f} us ({fl / t_plain / 1e6 / 157.3:.3f}) -> with planes {t_vpl:7.1f} us ({t_vpl - t_plain:+6.1f})   "
              f"weight-grad {t_old:7.1f} us -> plane GEMM {t_new:7.1f} us ({fl / t_new / 1e6 / 157.3:.3f} executed, {t_new - t_old:+7.1f})", flush=True)
    print(f"sum: forward {tf:.0f} -> {tv:.0f} us ({tv - tf:+.0f}), weight-grad {tw_old:.0f} -> {tw_new:.0f} us ({tw_new - tw_old:+.0f}); net {tv - tf + tw_new - tw_old:+.0f} us per step")


if __name__ == "__main__":
    main()
