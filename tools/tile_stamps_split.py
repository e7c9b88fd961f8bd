#!/usr/bin/env python3
"""Timeline of ONE workgroup of the split-operand GEMM (experiments build: make -C pytorch-camvid_amd/csrc experiments; s_memtime stamps of
waves 0 (group A) and 4 (group B)):   python tools/tile_stamps_split.py [fmt] [Cin] [Cout] [H] [W]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
exp = ctypes.CDLL(os.path.join(ROOT, "pytorch-camvid_amd", "lib", "libcvk_exp.so"))
fmt, ci, co, h, w = (int(v) for v in (sys.argv[1:6] + ["2", "256", "256", "90", "120"][len(sys.argv) - 1:]))
N, NX = 8, 64
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
T = N * ((h + 5) // 6) * ((w + 5) // 6)
Tp, Cp = (T + 255) // 256 * 256, (co + 127) // 128 * 128
dt = torch.bfloat16 if fmt == 3 else torch.float16
V = (torch.randn(NX * (ci // 32) * fmt * Tp * 32, device=dev) * 0.5).to(dt)
U = (torch.randn(NX * (ci // 32) * fmt * Cp * 32, device=dev) * 0.05).to(dt)
Mo = torch.empty(NX * T * co + 1024, device=dev)
am = torch.full((256,), 0x3F800000, device=dev, dtype=torch.int32)
f = exp.cvk_w2d_gemm_split_dbg
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
ntiles = NX * (Tp // 256) * (Cp // 128)
for wg in (0, ntiles // 2, ntiles - 300):
    dbg = torch.zeros(128, device=dev, dtype=torch.int64)
    for _ in range(3):
        assert f(fmt, 6, V.data_ptr(), U.data_ptr(), Mo.data_ptr(), am.data_ptr(), am.data_ptr(), NX, T, Tp, ci, co, Cp, dbg.data_ptr(), wg, s) == 0
    torch.cuda.synchronize()
    t = dbg.cpu().view(2, 64)
    ncs = ci // 32
    for g in range(2):
        r = t[g]
        n = int((r != 0).sum())
        rel = [(int(r[i]) - int(r[0])) for i in range(n)]
        # stamps: 0 entry, 1 first slice landed, then per slice: frag reads done, barrier passed (MFMA starts), MFMAs issued, DMA wait done; last two: loop end, stores issued
        print(f"wg {wg} group {'AB'[g]}: prologue {rel[1]}  | per slice (reads, barrier, mfma, wait): " +
              " ".join(f"[{rel[2+4*c]-rel[1+4*c] if c else rel[2]-rel[1]},{rel[3+4*c]-rel[2+4*c]},{rel[4+4*c]-rel[3+4*c]},{rel[5+4*c]-rel[4+4*c]}]" for c in range(ncs)) +
              f" | tail {rel[n-2]-rel[n-3]} epilogue {rel[n-1]-rel[n-2]}  total {rel[n-1]}")
