#!/usr/bin/env python3
"""Ablation timing of the fused F(4,3) kernel (csrc/wino4f.hip built with -DCVK_WINO4F_ABLATE into scratch/libcvk_abl.so):
which phase of the K step does the matrix pipe wait for?  Bits: 1 no pixel loads, 2 no transform/LDS stores, 4 no filter
DMA, 8 no epilogue, 16 no MFMAs."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_camvid_amd import _lib
lib = _lib.load()
abl = ctypes.CDLL(os.path.join(ROOT, "scratch", "libcvk_abl.so"))
f = abl.cvk_conv3x3_wino4f_ablate
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
N = 8
for (ci, co, H, W) in ((64, 64, 360, 480), (128, 64, 360, 480), (128, 128, 180, 240)):
    M = N * H * W
    x = torch.randn(M, ci, device="cuda"); w = torch.randn(co, 9 * ci, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    y = torch.empty(M, co, device="cuda")
    Uf = torch.empty(lib.cvk_wino4f_weight_floats(co, ci), device="cuda")
    _lib.check(lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), co, ci, 0, s))
    row = f"{ci}->{co} {H}x{W}:"
    variants = ((0, "full"), (1, "-loads"), (2, "-xform"), (4, "-dma"), (8, "-epi"), (9, "-loads-epi"), (11, "-loads-xform-epi"), (15, "mfma+lds only"),
                (16, "-mfma"), (31, "nothing"))
    best = {a: 1e9 for a, _ in variants}
    for rnd in range(4):                      # interleaved rounds, minimum per variant: the boxes' clocks drift by several %
        for a, name in variants:
            def run():
                rc = f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W, ci, co, co, a, s)
                assert rc == 0
            run(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            best[a] = min(best[a], e0.elapsed_time(e1) / 5 * 1e3)
    for a, name in variants:
        row += f"  {name} {best[a]:.0f}"
    print(row, flush=True)
