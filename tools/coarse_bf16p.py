"""Per-workgroup timeline of the ping-pong bf16 conv kernel (CVK_BF16P_DBG=16: s_memrealtime at start / after prologue /
after the K loop / at the end):  python tools/coarse_bf16p.py Cin Cout H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
nwg = N * ((h + 15) // 16) * ((w + 31) // 32) * ((co + 127) // 128)
st = torch.zeros(nwg * 4, device=dev, dtype=torch.int64)
for _ in range(3):
    check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), st.data_ptr(), st.data_ptr(), N, h, w, ci, co, co, s))
torch.cuda.synchronize()
t = st.cpu().view(nwg, 4).double() * 10e-3          # us (100 MHz ticks)
t0 = t[:, 0].min()
t = t - t0
print(f"{ci}->{co} @{h}x{w}: {nwg} workgroups, kernel span {float(t[:, 3].max()):.1f} us")
print(f"  prologue  mean {float((t[:,1]-t[:,0]).mean()):6.2f} us   (min {float((t[:,1]-t[:,0]).min()):.2f}, max {float((t[:,1]-t[:,0]).max()):.2f})")
print(f"  K loop    mean {float((t[:,2]-t[:,1]).mean()):6.2f} us   (min {float((t[:,2]-t[:,1]).min()):.2f}, max {float((t[:,2]-t[:,1]).max()):.2f})")
print(f"  epilogue  mean {float((t[:,3]-t[:,2]).mean()):6.2f} us   (min {float((t[:,3]-t[:,2]).min()):.2f}, max {float((t[:,3]-t[:,2]).max()):.2f})")
starts = t[:, 0].sort().values
print("  start times (us) every 64th workgroup in start order:", [round(float(v), 1) for v in starts[::64]])
