#!/usr/bin/env python3
"""Time the fused F(4,3) convolution of the 64/128-channel levels: exact-fp32 kernel (cvk_conv3x3_wino4f) against the opt-in fp16 split-operand
form (cvk_conv3x3_wino4h), headline shapes (batch 8).    python tools/study/wino4h_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_camvid_amd import _lib          # noqa: E402
from pytorch_camvid_amd._lib import check    # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream


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


print("layer (batch 8)                 exact fp32     fp16-split form   speed-up   (stats variants)")
for name, ci, co, h, w in [("down1.1", 64, 64, 360, 480), ("down2.0", 64, 128, 180, 240), ("down2.1", 128, 128, 180, 240), ("up4.0", 128, 64, 360, 480)]:
    N = 8
    x = torch.randn(N, h, w, ci, device=dev).clamp_min(0)
    wt = (torch.rand(co, 3, 3, ci, device=dev) * 2 - 1) / (9 * ci) ** 0.5
    b = torch.zeros(co, device=dev)
    nfl = lib.cvk_wino4f_weight_floats(co, ci)
    Uf = torch.empty(nfl, device=dev); Uh = torch.empty(nfl, device=dev)
    amw = torch.zeros(lib.cvk_amax_block_words(), device=dev, dtype=torch.int32); amx = torch.zeros_like(amw)
    check(lib.cvk_absmax_f32(wt.data_ptr(), wt.numel() // ci, ci, ci, amw.data_ptr(), s))
    check(lib.cvk_absmax_f32(x.data_ptr(), x.numel() // ci, ci, ci, amx.data_ptr(), s))
    check(lib.cvk_wino4f_weight_transform(wt.data_ptr(), Uf.data_ptr(), co, ci, 0, s))
    check(lib.cvk_wino4h_weight_transform(wt.data_ptr(), Uh.data_ptr(), amw.data_ptr(), co, ci, 0, s))
    y = torch.empty(N, h, w, co, device=dev)
    P = lib.cvk_wino4f_stat_partials(N, h, w)
    st = torch.zeros(2 * P * co + P, device=dev)
    t32 = timeit(lambda: check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, N, h, w, ci, co, co, 0, s)))
    th = timeit(lambda: check(lib.cvk_conv3x3_wino4h(x.data_ptr(), Uh.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, amx.data_ptr(), amw.data_ptr(), N, h, w, ci, co, co, 0, s)))
    t32s = timeit(lambda: check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co, N, h, w, ci, co, co, 0, s)))
    ths = timeit(lambda: check(lib.cvk_conv3x3_wino4h(x.data_ptr(), Uh.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co, amx.data_ptr(), amw.data_ptr(), N, h, w, ci, co, co, 0, s)))
    print(f"{name:8s} {ci:4d}->{co:4d} @{h}x{w}   {t32*1e6:8.1f} us   {th*1e6:8.1f} us      {t32/th:5.2f}x      {t32s*1e6:8.1f} / {ths*1e6:8.1f} us")
    del x, y, Uf, Uh
