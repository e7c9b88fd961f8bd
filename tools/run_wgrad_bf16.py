"""Run the bf16 weight-grad of one layer a few times — the target of tools/pmc_kernel.sh:  python tools/run_wgrad_bf16.py Cin Cout H W [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib = _lib.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
ci, co, h, w = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
N = 4
x = torch.randn(N, h, w, ci, device=dev).to(BF)
dy = torch.randn(N, h, w, co, device=dev).to(BF)
dw = torch.empty(co * 9 * ci, device=dev)
wsb = lib.cvk_conv3x3_wgrad_bf16s_workspace_bytes(N, h, w, ci, co)
ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
for _ in range(reps):
    check(lib.cvk_conv3x3_wgrad_bf16s(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, h, w, ci, ci, co, co, ws.data_ptr(), wsb, s))
torch.cuda.synchronize()
