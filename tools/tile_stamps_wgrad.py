"""Per-tile timeline of the row-stationary bf16 weight-grad (CVK_WGRAD_DBG=16): workgroup 0, waves 0 and 4.  python tools/tile_stamps_wgrad.py Cin Cout H W"""
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
dy = torch.randn(N, h, w, co, device=dev).to(BF)
dw = torch.empty(co * 9 * ci, device=dev)
wsb = lib.cvk_conv3x3_wgrad_bf16s_workspace_bytes(N, h, w, ci, co)
ws = torch.zeros(wsb + (1 << 16), device=dev, dtype=torch.uint8)
for _ in range(3):
    check(lib.cvk_conv3x3_wgrad_bf16s(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, h, w, ci, ci, co, co, ws.data_ptr(), wsb, s))
torch.cuda.synchronize()
t = ws[wsb: wsb + 2 * 24 * 4 * 8].view(torch.int64).view(2, 24, 4).cpu()
for g in (0, 1):
    print("wave", 4 * g)
    for i in range(3, 11):
        r = t[g, i]; nxt = t[g, i + 1][0]
        print(f"  tile {i}: rows+DMA {int(r[1]-r[0]):6d}  vmcnt wait {int(r[2]-r[1]):5d}  barrier {int(r[3]-r[2]):5d}  -> next tile {int(nxt-r[3]):4d}   total {int(nxt-r[0]):6d} cycles (72 MFMAs x 2 waves = 4608 matrix cycles)")
k = t[0, 23]
cyc, ns = int(k[1] - k[0]), int(k[3] - k[2]) * 10
ntile = -(-(N * -(-h // 8) * -(-w // 32)) // max(1, -(-256 // (-(-ci // 64) * -(-co // 64)))))
print(f"workgroup 0: {cyc} shader cycles in {ns / 1e3:.1f} us = {cyc / ns:.2f} GHz; ~{ntile} tiles -> {cyc / ntile:.0f} cycles per tile including prologue, exchange and slab write")
first, last = t[0, 0], t[0, 22]
print(f"  before the first tile {int(first[0] - k[0])} cycles; tiles 0-22 {int(last[3] - first[0])} cycles")
if int(os.environ.get("CVK_WGRAD_DBG", "0")) & 64:
    r = ws[wsb + 2 * 24 * 4 * 8: wsb + 2 * 24 * 4 * 8 + 2 * 24 * 4 * 8].view(torch.int64).cpu()
    for g in (0, 1):
        v = r[g * 24 * 4 + g * 8: g * 24 * 4 + g * 8 + 11].tolist()
        print(f"wave {4 * g}, tile 9: halo-row start cycles {v[:10]}  end {v[10]}")
        print("   row durations", [v[i + 1] - v[i] for i in range(10)])
