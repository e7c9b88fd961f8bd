"""Per-tile timeline of workgroup 0 of the 64-channel ping-pong kernel (CVK_BF16H_DBG=1): python tools/tile_stamps_h.py Cin Cout H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _exp  # noqa: F401  (experiments build of the library: CVK_* knobs and time stamps)
import torch
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib = _lib.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
ci, co, h, w = (int(v) for v in sys.argv[1:5])
N = 4
x = torch.randn(N, h, w, ci, device=dev).to(BF)
wt = torch.randn(co, 3, 3, ci, device=dev) * 0.05
wp = torch.empty(lib.cvk_bf16s_rows_pad(co) * 9 * ci, device=dev, dtype=BF)
check(lib.cvk_pack_weight_fwd_bf16(wt.data_ptr(), wp.data_ptr(), co, ci, ci, s))
y = torch.empty(N * h * w * co, device=dev, dtype=BF)
with_stats = os.environ.get("CVK_BF16H_DBG") == "2"          # 2: the statistics epilogue (stamps behind the counts), 1: without
if with_stats:
    P = lib.cvk_bf16s_stat_partials_c(N, h, w, ci, co)
    buf = torch.zeros(2 * P * co + ((P + 1) & ~1) + 2 * (16 * 8 + 64), device=dev, dtype=torch.float32)
    bias = torch.zeros(co, device=dev)
    cntp = buf.data_ptr() + 4 * 2 * P * co
    for _ in range(3):
        check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(), buf.data_ptr(), cntp, N, h, w, ci, co, co, s))
    torch.cuda.synchronize()
    st = buf[2 * P * co + ((P + 1) & ~1):].view(torch.int64)
else:
    st = torch.zeros(16 * 8 + 64, device=dev, dtype=torch.int64)
    for _ in range(3):
        check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), st.data_ptr(), st.data_ptr(), N, h, w, ci, co, co, s))
    torch.cuda.synchronize()
t = st[:128].cpu().view(16, 8).double() * 10e-3
names = ["top wait+barrier", "K loop", "drain", "next prologue issue", "compute+stage", "store+stats"]
for i in range(2, 10):
    r = t[i]
    d = [float(r[j + 1] - r[j]) for j in range(6)]
    nxt = float(t[i + 1][0] - r[0])
    print(f"tile {i}: " + "  ".join(f"{n} {v:5.2f}" for n, v in zip(names, d)) + f"   total {nxt:5.2f} us")
