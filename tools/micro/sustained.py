#!/usr/bin/env python3
"""Does the F(4,3) GEMM slow down when it runs back to back for a long time (power / clock management) rather than in a
5-launch burst?  One mid-size layer (512->256 at 8x90x120), bursts of 5 vs 400 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib = _lib.load(); s = torch.cuda.current_stream().cuda_stream
N, H, W, ci, co = 8, 90, 120, 512, 256
M = N * H * W
x = torch.randn(M, ci, device="cuda"); w = torch.randn(co, 9 * ci, device="cuda") * 0.05
U = torch.empty(6 * co * 3 * ci, device="cuda")
check(lib.cvk_wino4_weight_transform(w.data_ptr(), U.data_ptr(), co, ci, s))
ws = torch.empty(lib.cvk_conv3x3_wino4_workspace_bytes(N, H, W, ci, co), dtype=torch.uint8, device="cuda")
def run(n):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        check(lib.cvk_conv3x3_wino4_gemm(x.data_ptr(), U.data_ptr(), ws.data_ptr(), N, H, W, ci, co, co, s))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
run(3)
for n in (5, 5, 50, 400, 400, 5):
    t = run(n); print(f"{n:4d} launches back to back: {t:7.1f} us/launch  {18.0 * M * ci * co / t / 1e6:6.1f} TFLOP/s algorithmic")
    torch.cuda.synchronize()
