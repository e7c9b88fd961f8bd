"""Phase timeline of the ping-pong bf16 conv kernel (CVK_BF16P_DBG=8|...: s_memtime stamps of workgroup 0, first 48 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib = _lib.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
N, h, w, ci, co = 4, 90, 120, 1024, 512
x = torch.randn(N, h, w, ci, device=dev).to(BF)
wt = torch.randn(co, 3, 3, ci, device=dev) * 0.05
wp = torch.empty(lib.cvk_bf16s_rows_pad(co) * 9 * ci, device=dev, dtype=BF)
check(lib.cvk_pack_weight_fwd_bf16(wt.data_ptr(), wp.data_ptr(), co, ci, ci, s))
y = torch.empty(N * h * w * co, device=dev, dtype=BF)
st = torch.zeros(8 * 48 * 8 + 1024, device=dev, dtype=torch.int32)
for _ in range(3):
    check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), st.data_ptr(), st.data_ptr(), N, h, w, ci, co, co, s))
torch.cuda.synchronize()
t = st[: 8 * 48 * 8].cpu().view(8, 48, 8).long()
names = ["dma", "reads+lgkm", "vmwait", "barrier", "mfma", "barrier2"]
for wv in (0, 4):
    print("wave", wv)
    for stp in range(20, 30):
        r = t[wv, stp]; nxt = t[wv, stp + 1][0]
        d = [int(r[1] - r[0]), int(r[2] - r[1]), int(r[3] - r[2]), int(r[4] - r[3]), int(r[5] - r[4]), int(nxt - r[5])]
        print(f"  step {stp}: " + "  ".join(f"{n} {v:5d}" for n, v in zip(names, d)) + f"   total {int(nxt - r[0])}")
# memtime ticks at 100 MHz? print raw unit check
print("ticks per step (wave0):", int(t[0, 40, 0] - t[0, 20, 0]) / 20)
