"""Raw C-ABI parity of the thin-channel kernels (csrc/thin.hip) against the fp64 operator they replace: the stem
nn.Conv2d(3, 64, 3, padding=1) (/root/reference/models/unet.py:103) and the classifier head nn.Conv2d(64, class_num, 3, padding=1)
(models/unet.py:127) — forward with the BatchNorm batch statistics (unet.py:12), data-grad and weight-grad (train.py:131).
Plain fp32 multiply-accumulate (no Winograd): 1e-6 relative L2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from pytorch_camvid_amd import _lib
    return _lib.load(), _lib.check


def _stream():
    return torch.cuda.current_stream().cuda_stream


GEOM = [  # N, H, W — ragged widths (W % 16, W % 4 != 0), fewer rows than one chunk, several chunks per image
    (1, 5, 7),
    (2, 9, 13),
    (1, 24, 33),
    (2, 45, 60),
    (3, 64, 96),
    (2, 1, 1),
    (1, 130, 50),
]


def _fwd(x_nhwc, w_oihw, bias, ldy, stats=True):
    lib, check = _lib()
    N, H, W, ld = x_nhwc.shape
    Cout, Cin = w_oihw.shape[:2]
    wk = torch.zeros(Cout, 9, ld, device="cuda")
    wk[:, :, :Cin] = w_oihw.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    assert lib.cvk_thin_fwd_supported(ld, Cout, ldy) == 1
    y = torch.full((N, H, W, ldy), float("nan"), device="cuda")
    P = lib.cvk_thin_stat_partials(N, H, W, ld)
    st = torch.full((2 * P * Cout + P,), float("nan"), device="cuda") if stats else None
    check(lib.cvk_conv3x3_thin_fwd(x_nhwc.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                   st.data_ptr() if stats else None, st.data_ptr() + 8 * P * Cout if stats else None,
                                   N, H, W, ld, Cout, ldy, _stream()), "thin fwd")
    torch.cuda.synchronize()
    return y, st, P


def _check_stats(st, P, C, ref):
    M = ref.shape[0] * ref.shape[2] * ref.shape[3]
    sums = st[:P * C].view(P, C).double().cpu()
    m2 = st[P * C:2 * P * C].view(P, C).double().cpu()
    cnt = st[2 * P * C:].double().cpu()
    assert int(cnt.sum().item()) == M
    mean = sums.sum(0) / M
    var = (m2.sum(0) + (cnt[:, None] * (sums / cnt[:, None] - mean) ** 2).sum(0)) / M
    rmean, rvar = ref.mean(dim=(0, 2, 3)), ref.var(dim=(0, 2, 3), unbiased=False)
    assert (mean - rmean).abs().max().item() < 2e-6 * max(1.0, rmean.abs().max().item())
    # fp32 sums: the variance is exact to 2e-5 of itself plus the rounding of the squared values it is a difference of
    assert ((var - rvar).abs() <= 2e-5 * rvar + 2e-7 * (ref ** 2).mean(dim=(0, 2, 3))).all()


@pytest.mark.parametrize("N,H,W", GEOM)
@pytest.mark.parametrize("Cout,ldy", [(12, 12), (16, 16), (5, 8)])
def test_head_forward_and_statistics_vs_fp64(N, H, W, Cout, ldy):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + Cout)
    x = torch.randn(N, 64, H, W, generator=g)
    w = torch.randn(Cout, 64, 3, 3, generator=g) * (2.0 / (9 * 64)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y, st, P = _fwd(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), b.cuda(), ldy)
    got = y[..., :Cout].permute(0, 3, 1, 2).double().cpu()
    assert torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-6, rel
    if ldy > Cout:
        assert (y[..., Cout:] == 0).all()                                 # padding columns carry zeros
    _check_stats(st, P, Cout, ref)
    # without statistics / without bias: same values
    y2, _, _ = _fwd(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), b.cuda(), ldy, stats=False)
    assert torch.equal(y2[..., :Cout], y[..., :Cout])


@pytest.mark.parametrize("N,H,W", GEOM)
@pytest.mark.parametrize("Cout,ld_dy", [(12, 12), (16, 16), (5, 8)])
def test_head_weight_grad_vs_fp64(N, H, W, Cout, ld_dy):
    lib, check = _lib()
    g = torch.Generator().manual_seed(N * 77 + H + W + Cout)
    x = torch.randn(N, 64, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    xd = x.double().requires_grad_(False)
    wref = torch.zeros(Cout, 64, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, wref, padding=1).backward(dy.double())
    ref = wref.grad.permute(0, 2, 3, 1)                                   # [Cout][3][3][64]
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyn = torch.zeros(N, H, W, ld_dy, device="cuda")
    dyn[..., :Cout] = dy.permute(0, 2, 3, 1).cuda()
    assert lib.cvk_thin_wgrad_supported(64, 64, Cout, ld_dy) == 1
    wsb = lib.cvk_conv3x3_thin_wgrad_workspace_bytes(N, H, W, 64, Cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((Cout, 3, 3, 64), float("nan"), device="cuda")
        check(lib.cvk_conv3x3_thin_wgrad(xn.data_ptr(), dyn.data_ptr(), dw.data_ptr(), N, H, W, 64, 64, Cout, ld_dy, ws.data_ptr(), wsb,
                                         _stream()), "thin wgrad")
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])                                  # fixed-order reduction
    got = outs[0].double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-6, rel


@pytest.mark.parametrize("N,H,W", GEOM)
@pytest.mark.parametrize("Cin,ld,Cout", [(3, 4, 64), (3, 4, 128), (5, 8, 64), (3, 4, 36)])
def test_stem_forward_and_statistics_vs_fp64(N, H, W, Cin, ld, Cout):
    g = torch.Generator().manual_seed(N * 31 + H * 7 + W + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xn = torch.zeros(N, H, W, ld, device="cuda")
    xn[..., :Cin] = x.permute(0, 2, 3, 1).cuda()
    y, st, P = _fwd(xn, w.cuda(), b.cuda(), Cout)
    got = y.permute(0, 3, 1, 2).double().cpu()
    assert torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-6, rel
    _check_stats(st, P, Cout, ref)


@pytest.mark.parametrize("N,H,W", GEOM)
def test_head_data_grad_vs_fp64(N, H, W):
    """dx = conv_transpose(dy, w): the same kernel on the rotated / transposed pack [Cin][9][Cout_ld = 12], no bias, no statistics."""
    lib, check = _lib()
    Cout, Cin = 12, 64
    g = torch.Generator().manual_seed(N + H + W)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    dy = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9, Cout).contiguous().cuda()        # [ci][tap'][co] = w[co][ci][8 - tap']
    dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    check(lib.cvk_conv3x3_thin_fwd(dyn.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, None, N, H, W, 12, Cin, Cin, _stream()), "thin dgrad")
    torch.cuda.synchronize()
    got = dx.permute(0, 3, 1, 2).double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 1e-6, rel


@pytest.mark.parametrize("N,H,W", GEOM)
@pytest.mark.parametrize("Cin,Cout", [(3, 64), (3, 128), (4, 64), (1, 36)])
def test_stem_weight_grad_vs_fp64(N, H, W, Cin, Cout):
    lib, check = _lib()
    g = torch.Generator().manual_seed(N * 5 + H + W + Cout + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wref, padding=1).backward(dy.double())
    ref = wref.grad.permute(0, 2, 3, 1)                                   # [Cout][3][3][Cin]
    xn = torch.zeros(N, H, W, 4, device="cuda")
    xn[..., :Cin] = x.permute(0, 2, 3, 1).cuda()
    dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
    assert lib.cvk_thin_wgrad_supported(Cin, 4, Cout, Cout) == 1
    wsb = lib.cvk_conv3x3_thin_wgrad_workspace_bytes(N, H, W, 4, Cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
        check(lib.cvk_conv3x3_thin_wgrad(xn.data_ptr(), dyn.data_ptr(), dw.data_ptr(), N, H, W, Cin, 4, Cout, Cout, ws.data_ptr(), wsb, _stream()),
              "thin wgrad")
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    got = outs[0].double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-6, rel
