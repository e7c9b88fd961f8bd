"""Timing of the bf16 forward / data-grad conv kernels on a few configs[3] layers (no statistics), for A/B runs under
the experiment knobs CVK_BF16P / CVK_BF16P_H64 / CVK_BF16P_H128 / CVK_STREAM_HINTS (experiments build):   python tools/bench_bf16p.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _exp  # noqa: F401  (experiments build of the library: CVK_* knobs and time stamps)
import torch
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

lib = _lib.load()
dev = torch.device("cuda:0")
BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
N = 4
LAYERS = [("128->128@360", 128, 128, 360, 480), ("256->128@360", 256, 128, 360, 480), ("256->256@180", 256, 256, 180, 240),
          ("512->512@90", 512, 512, 90, 120), ("1024->512@90", 1024, 512, 90, 120), ("1024->1024@45", 1024, 1024, 45, 60)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


tag = sys.argv[1] if len(sys.argv) > 1 else ""
out = []
for name, ci, co, h, w in LAYERS:
    x = torch.randn(N, h, w, ci, device=dev).to(BF)
    wt = torch.randn(co, 3, 3, ci, device=dev) * 0.05
    wp = torch.empty(lib.cvk_bf16s_rows_pad(co) * 9 * ci, device=dev, dtype=BF)
    check(lib.cvk_pack_weight_fwd_bf16(wt.data_ptr(), wp.data_ptr(), co, ci, ci, s))
    y = torch.empty(N * h * w * co, device=dev, dtype=BF)
    flops = 18.0 * N * h * w * ci * co
    t = timeit(lambda: check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, None, N, h, w, ci, co, co, s)))
    out.append(f"{name} {t*1e6:7.1f}us {flops/t/1e12:6.0f}TF")
print(f"{tag:14s}", " | ".join(out), flush=True)
