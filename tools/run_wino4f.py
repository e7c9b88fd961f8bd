#!/usr/bin/env python3
"""One layer through the fused F(4,3) kernel, a few launches (target of rocprofv3 --pmc runs: tools/pmc_wino4f.sh).
usage: run_wino4f.py Cin Cout H W [N] [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
lib = _lib.load(); check = _lib.check
ci, co, H, W = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 8
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 4
s = torch.cuda.current_stream().cuda_stream
M = N * H * W
x = torch.randn(M, ci, device="cuda"); w = torch.randn(co, 9 * ci, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
y = torch.empty(M, co, device="cuda")
Uf = torch.empty(lib.cvk_wino4f_weight_floats(co, ci), device="cuda")
check(lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), co, ci, 0, s))
P = lib.cvk_wino4f_stat_partials(N, H, W); st = torch.zeros(2 * P * co + P, device="cuda")
for _ in range(reps):
    check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co, N, H, W, ci, co, co, 0, s))
torch.cuda.synchronize()
print("done", float(y[0, 0]))
