"""OPT-IN fp16 split-operand form of the fused F(4,3) convolution (csrc/wino4f.hip k_conv3x3_wino4f<.., H2>; runner.w2d_split = 2 — never the
default): nn.Conv2d(cin, cout, 3, padding=1) forward (models/unet.py:11) and its data-grad for the 64/128-channel levels, against an fp64
convolution, the exact-fp32 kernel beside it; BatchNorm statistics and the BatchNorm-backward sums of the epilogue against the fp32 kernel's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from pytorch_camvid_amd import _lib
    return _lib.load(), _lib.check


def _block(lib, dev):
    return torch.zeros(lib.cvk_amax_block_words(), device=dev, dtype=torch.int32)


def _amax(lib, check, t, s):
    a = _block(lib, t.device)
    C = t.shape[-1]
    check(lib.cvk_absmax_f32(t.data_ptr(), t.numel() // C, C, C, a.data_ptr(), s), "cvk_absmax_f32")
    return a


def _both(lib, check, x, w, bias, dgrad, s, stats=False):
    """conv through the fp16 form and through the exact-fp32 kernel; w is the forward filter [Cout][3][3][Cin] (dgrad: x has Cout channels)"""
    dev = x.device
    N, H, W, Ck = x.shape
    Cn = w.shape[3] if dgrad else w.shape[0]
    nfl = lib.cvk_wino4f_weight_floats(Cn, Ck)
    Uf = torch.empty(nfl, device=dev); Uh = torch.empty(nfl, device=dev)
    amw, amx = _amax(lib, check, w, s), _amax(lib, check, x, s)
    check(lib.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), Cn, Ck, dgrad, s), "weight")
    check(lib.cvk_wino4h_weight_transform(w.data_ptr(), Uh.data_ptr(), amw.data_ptr(), Cn, Ck, dgrad, s), "weight h")
    y32 = torch.full((N, H, W, Cn), float("nan"), device=dev); yh = torch.full((N, H, W, Cn), float("nan"), device=dev)
    P = lib.cvk_wino4f_stat_partials(N, H, W)
    st32 = torch.zeros(2 * P * Cn + P, device=dev) if stats else None
    sth = torch.zeros(2 * P * Cn + P, device=dev) if stats else None
    bp = bias.data_ptr() if bias is not None else None
    check(lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), bp, y32.data_ptr(), st32.data_ptr() if stats else None,
                                 st32.data_ptr() + 4 * 2 * P * Cn if stats else None, N, H, W, Ck, Cn, Cn, 0, s), "wino4f")
    check(lib.cvk_conv3x3_wino4h(x.data_ptr(), Uh.data_ptr(), bp, yh.data_ptr(), sth.data_ptr() if stats else None,
                                 sth.data_ptr() + 4 * 2 * P * Cn if stats else None, amx.data_ptr(), amw.data_ptr(), N, H, W, Ck, Cn, Cn, 0, s), "wino4h")
    return yh, y32, sth, st32


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 37, 50, 64, 64), (1, 24, 131, 128, 64), (2, 20, 36, 64, 128), (1, 9, 7, 32, 96)])
@pytest.mark.parametrize("mag", [1.0, 1e-20, 1e20])
def test_forward_vs_fp64(N, H, W, Cin, Cout, mag):
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(H * W)
    x = (torch.randn(N, H, W, Cin, generator=g).clamp_min(0) * mag).to(dev)
    w = ((torch.rand(Cout, 3, 3, Cin, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev) * 0.1 * mag
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    yh, y32, sth, st32 = _both(lib, check, x, w, b, 0, s, stats=True)
    assert torch.isfinite(yh).all()
    eh = ((yh.double() - ref).norm() / ref.norm()).item(); e32 = ((y32.double() - ref).norm() / ref.norm()).item()
    print(f"{N}x{H}x{W} {Cin}->{Cout} x{mag:g}: relative L2 vs fp64: exact-fp32 kernel {e32:.2e}, fp16-split form {eh:.2e}")
    assert eh <= 1.35 * e32 + 2e-8 and eh < 2e-6, (eh, e32)
    # the statistics partials of the epilogue (sum, M2, count) agree with the fp32 kernel's to rounding
    P = lib.cvk_wino4f_stat_partials(N, H, W)
    a, b_ = sth[:P * Cout], st32[:P * Cout]
    assert ((a - b_).abs().max() / b_.abs().max()).item() < 1e-5
    assert torch.equal(sth[2 * P * Cout:], st32[2 * P * Cout:])                       # pixel counts
    if mag <= 1.0:                                                                    # (the squares of 1e20-sized values leave fp32 in both kernels)
        m2h, m232 = sth[P * Cout:2 * P * Cout], st32[P * Cout:2 * P * Cout]
        assert ((m2h - m232).abs().max() / m232.abs().max()).item() < 1e-4


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 37, 50, 64, 64), (1, 24, 131, 64, 128), (2, 20, 36, 128, 64)])
def test_data_grad_vs_fp64(N, H, W, Cin, Cout):
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(H + W)
    w = ((torch.rand(Cout, 3, 3, Cin, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev)
    dy = (torch.randn(N, H, W, Cout, generator=g) * torch.exp(1.5 * torch.randn(N, H, W, Cout, generator=g)) * 1e-6).to(dev)
    ref = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    dxh, dx32, _, _ = _both(lib, check, dy, w, None, 1, s)
    eh = ((dxh.double() - ref).norm() / ref.norm()).item(); e32 = ((dx32.double() - ref).norm() / ref.norm()).item()
    print(f"dgrad {N}x{H}x{W} {Cout}->{Cin}: relative L2 vs fp64: exact-fp32 kernel {e32:.2e}, fp16-split form {eh:.2e}")
    assert torch.isfinite(dxh).all() and eh <= 1.35 * e32 + 2e-8 and eh < 2e-6, (eh, e32)
    # bitwise reproducible
    dxh2, _, _, _ = _both(lib, check, dy, w, None, 1, s)
    assert torch.equal(dxh, dxh2)
