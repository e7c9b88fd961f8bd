"""Run the bf16 forward conv (no statistics) of one layer a few times — the target of tools/pmc_kernel.sh.
    python tools/run_bf16p.py Cin Cout H W [reps] [stats]"""
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
with_stats = len(sys.argv) > 6
N = 4
x = torch.randn(N, h, w, ci, device=dev).to(BF)
wt = torch.randn(co, 3, 3, ci, device=dev) * 0.05
b = torch.zeros(co, device=dev)
wp = torch.empty(lib.cvk_bf16s_rows_pad(co) * 9 * ci, device=dev, dtype=BF)
check(lib.cvk_pack_weight_fwd_bf16(wt.data_ptr(), wp.data_ptr(), co, ci, ci, s))
y = torch.empty(N * h * w * co, device=dev, dtype=BF)
P = lib.cvk_bf16s_stat_partials_c(N, h, w, ci, co)
st = torch.empty(2 * P * co + P, device=dev)
for _ in range(reps):
    if with_stats:
        check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co, N, h, w, ci, co, co, s))
    else:
        check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, None, N, h, w, ci, co, co, s))
torch.cuda.synchronize()
