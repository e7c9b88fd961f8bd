#!/usr/bin/env python3
"""Bit-pattern fixture of the fused F(4,3) convolution (csrc/wino4f.hip) — runs on a GPU box.

    python tests/golden/make_wino4f_bits.py [path/to/libcvk.so] [out.npz]

Commits eb2087d / 36c1d00 rewrote the input transform of k_conv3x3_wino4f on channel pairs with packed FMAs and claimed "bitwise the same
values" in their messages.  This script runs ONE shape through a given build of the library (raw C ABI, ctypes) and stores, for the output
tensor and the BatchNorm statistics partials, 1024 wrapping uint32 sums of the IEEE bit patterns (element i goes to bucket i % 1024): 4 KiB
per tensor, any changed bit changes a bucket.  tests/golden/wino4f_bits.npz was written from the library built at bf80833 (the commit
BEFORE eb2087d); tests/test_gpu_wino4f.py::test_fused_f43_bitwise_equals_the_pre_packed_kernel holds today's kernel to it.
Inputs are made by recipe (torch CPU generators), like every other fixture here."""
import ctypes
import os
import sys

import numpy as np
import torch

SHAPE = dict(N=2, H=24, W=44, Cin=64, Cout=64)      # 11 column groups per row (ragged against the 128-group tiles), 2 x 24 rows


def inputs():
    g = torch.Generator().manual_seed(4343)
    s = SHAPE
    x = torch.randn(s["N"] * s["H"] * s["W"], s["Cin"], generator=g)
    w = torch.randn(s["Cout"], 9 * s["Cin"], generator=g) * 0.05
    b = torch.randn(s["Cout"], generator=g)
    return x, w, b


def fold(t):
    """1024 wrapping uint32 sums of the bit patterns."""
    a = t.detach().cpu().contiguous().view(torch.int32).numpy().view(np.uint32).ravel()
    pad = (-a.size) % 1024
    a = np.concatenate([a, np.zeros(pad, np.uint32)]).reshape(-1, 1024).astype(np.uint64)
    return (a.sum(axis=0) & 0xFFFFFFFF).astype(np.uint32)


def run(lib):
    s = SHAPE
    dev = torch.device("cuda:0")
    x, w, b = (t.to(dev) for t in inputs())
    st = torch.cuda.current_stream().cuda_stream
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.cvk_wino4f_weight_floats.restype = ctypes.c_size_t
    lib.cvk_wino4f_weight_floats.argtypes = [ci, ci]
    lib.cvk_wino4f_weight_transform.argtypes = [vp, vp, ci, ci, ci, vp]
    lib.cvk_wino4f_stat_partials.argtypes = [ci, ci, ci]
    lib.cvk_conv3x3_wino4f.argtypes = [vp] * 6 + [ci] * 7 + [vp]
    out = {}
    for tag, dgrad in (("fwd", 0), ("dgrad", 1)):
        Uf = torch.empty(lib.cvk_wino4f_weight_floats(s["Cout"], s["Cin"]), device=dev)
        assert lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), s["Cout"], s["Cin"], dgrad, st) == 0
        y = torch.zeros(s["N"] * s["H"] * s["W"], s["Cout"], device=dev)
        P = lib.cvk_wino4f_stat_partials(s["N"], s["H"], s["W"])
        stats = torch.zeros(2 * P * s["Cout"] + P, device=dev)
        if dgrad:       # the data-grad form: no bias, no statistics
            rc = lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), None, y.data_ptr(), None, None, s["N"], s["H"], s["W"], s["Cin"], s["Cout"],
                                        s["Cout"], 0, st)
        else:
            rc = lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                        stats.data_ptr() + 8 * P * s["Cout"], s["N"], s["H"], s["W"], s["Cin"], s["Cout"], s["Cout"], 0, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        out[tag + "_y"] = fold(y)
        out[tag + "_y_first"] = y.flatten()[:16].cpu().numpy()
        if not dgrad:
            out["fwd_stats"] = fold(stats)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(here)), "pytorch-camvid_amd", "lib", "libcvk.so")
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "wino4f_bits.npz")
    res = run(ctypes.CDLL(path))
    np.savez_compressed(dst, library=os.path.basename(path), **res)
    print("wrote", dst, {k: (v[:2] if hasattr(v, "__len__") else v) for k, v in res.items()})
