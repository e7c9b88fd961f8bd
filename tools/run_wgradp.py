#!/usr/bin/env python3
"""One layer through the planes-based F(4,3) weight-grad, a few launches (rocprofv3 target).  usage: run_wgradp.py Cin Cout H W [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
lib = _lib.load(); check = _lib.check
ci, co, H, W = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 8
s = torch.cuda.current_stream().cuda_stream
M = N * H * W
x = torch.randn(M, ci, device="cuda"); dy = torch.randn(M, co, device="cuda"); dw = torch.empty(co, 9 * ci, device="cuda")
wsb = lib.cvk_conv3x3_wgradp_workspace_bytes(N, H, W, ci, co); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for _ in range(4):
    check(lib.cvk_conv3x3_wgradp(x.data_ptr(), dy.data_ptr(), None, dw.data_ptr(), N, H, W, ci, ci, co, co, ws.data_ptr(), wsb, s))
torch.cuda.synchronize()
print("done")
