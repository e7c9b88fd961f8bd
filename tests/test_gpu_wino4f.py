"""Raw C-ABI parity of the FUSED 1-D Winograd F(4,3) kernels (csrc/wino4f.hip) against the fp64 operator they replace:
nn.Conv2d(cin, cout, 3, padding=1) of /root/reference/models/unet.py:11 (forward, + the BatchNorm batch statistics the
block's train-mode BN needs, unet.py:12) and its data-gradient (backward of train.py:131).  Tolerances: the F(4,3)
transform constants (4, 5, 8) round 6-8e-7 relative L2 per layer (tests/test_drift_cpu.py); stated below per check."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from pytorch_camvid_amd import _lib
    return _lib.load(), _lib.check


def _fused_conv(x_nhwc, w_oihw, bias, stats=True, dgrad=False, max_wg=0):
    """x [N,H,W,Cin] fp32 cuda; w OIHW (forward filter).  dgrad: x is dy [N,H,W,Cout_f], result is dx [N,H,W,Cin_f]."""
    lib, check = _lib()
    s = torch.cuda.current_stream().cuda_stream
    N, H, W, Ck = x_nhwc.shape
    wcl = w_oihw.permute(0, 2, 3, 1).contiguous()          # [Cout][3][3][Cin]
    Cn = w_oihw.shape[1] if dgrad else w_oihw.shape[0]
    assert Ck == (w_oihw.shape[0] if dgrad else w_oihw.shape[1])
    Uf = torch.empty(lib.cvk_wino4f_weight_floats(Cn, Ck), device="cuda")
    check(lib.cvk_wino4f_weight_transform(wcl.data_ptr(), Uf.data_ptr(), Cn, Ck, 1 if dgrad else 0, s), "weight")
    y = torch.full((N, H, W, Cn), float("nan"), device="cuda")
    P = lib.cvk_wino4f_stat_partials(N, H, W)
    st = torch.zeros(2 * P * Cn + P, device="cuda") if stats else None
    check(lib.cvk_conv3x3_wino4f(x_nhwc.data_ptr(), Uf.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                 st.data_ptr() if stats else None, st.data_ptr() + 4 * 2 * P * Cn if stats else None,
                                 N, H, W, Ck, Cn, Cn, max_wg, s), "conv")
    torch.cuda.synchronize()
    return y, st, P


CASES = [  # N, H, W, Cin, Cout  — ragged widths (W % 4 != 0), rows < one tile, several tiles per workgroup, 1-3 n-tiles
    (1, 5, 7, 32, 64),
    (2, 9, 13, 64, 64),
    (1, 16, 36, 64, 128),
    (2, 33, 50, 128, 64),
    (1, 45, 60, 128, 136),
    (3, 64, 96, 64, 64),
    (8, 90, 120, 64, 64),      # 21600 tile rows = 169 tiles: partial last tile
    (2, 3, 4, 512, 512),       # one column group per image row (Wt = 1), deep K: the SegNet bottleneck geometries
    (2, 6, 8, 512, 256),
    (2, 12, 16, 256, 256),
    (1, 2, 3, 64, 64),         # W < 4
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", CASES)
def test_fused_forward_and_statistics_vs_fp64(N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)                    # the reference operator, fp64
    y, st, P = _fused_conv(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), b.cuda())
    got = y.permute(0, 3, 1, 2).double().cpu()
    assert torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 3e-6, rel                                                           # F(4,3) rounding: ~7e-7 per layer
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # statistics partials: [sum | M2 about the partial mean] + counts  ->  batch mean / biased variance (unet.py:12)
    M = N * H * W
    sums = st[:P * Cout].view(P, Cout).double().cpu()
    m2 = st[P * Cout:2 * P * Cout].view(P, Cout).double().cpu()
    cnt = st[2 * P * Cout:].double().cpu()
    assert int(cnt.sum().item()) == M
    mean = sums.sum(0) / M
    var = (m2.sum(0) + (cnt[:, None] * (sums / cnt[:, None] - mean) ** 2).sum(0)) / M
    rmean, rvar = ref.mean(dim=(0, 2, 3)), ref.var(dim=(0, 2, 3), unbiased=False)
    assert (mean - rmean).abs().max().item() < 2e-6 * max(1.0, rmean.abs().max().item())
    assert ((var - rvar).abs() / rvar).max().item() < 2e-5


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 13, 64, 64), (1, 16, 36, 64, 128), (2, 33, 50, 128, 64), (3, 64, 96, 64, 64)])
def test_fused_data_grad_vs_fp64(N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(7 + Cin + Cout)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    dy = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)                     # d/dx of conv2d(x, w, padding=1)
    dx, _, _ = _fused_conv(dy.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), None, stats=False, dgrad=True)
    got = dx.permute(0, 3, 1, 2).double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 3e-6, rel


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 13, 64, 64), (1, 16, 36, 64, 128), (2, 33, 50, 128, 64), (3, 64, 96, 64, 64),
                                             (1, 2, 3, 64, 64)])
def test_fused_data_grad_leaves_the_producers_batchnorm_backward_sums(N, H, W, Cin, Cout):
    """cvk_conv3x3_wino4f_bnred: the data-grad dx is bitwise the plain launch's, and its epilogue leaves the two column sums
    the producer block's BatchNorm+ReLU backward (reference models/unet.py:12-13) starts with: sum g and sum g * xhat,
    g = dx where the producer's ReLU passed, xhat = (yP - mean) * rstd — checked against fp64 of the same dx (fp32 partial
    sums over <= 512 values per tile and channel, fp64 across tiles: 2e-5 of the sums' scale)."""
    lib, check = _lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11 + Cin + Cout + W)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    dy = torch.randn(N, H, W, Cout, generator=g).cuda()
    yP = (torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.3).cuda()
    gamma, beta = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    mean = yP.mean(dim=(0, 1, 2))
    rstd = (yP.var(dim=(0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    scale = gamma * rstd
    shift = beta - mean * scale
    plain, _, _ = _fused_conv(dy, w.cuda(), None, stats=False, dgrad=True)
    wcl = w.permute(0, 2, 3, 1).contiguous().cuda()
    Uf = torch.empty(lib.cvk_wino4f_weight_floats(Cin, Cout), device="cuda")
    check(lib.cvk_wino4f_weight_transform(wcl.data_ptr(), Uf.data_ptr(), Cin, Cout, 1, s), "weight")
    P = lib.cvk_wino4f_stat_partials(N, H, W)
    for cap in (0, 3):
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        part = torch.full((2 * P * Cin,), float("nan"), device="cuda")
        check(lib.cvk_conv3x3_wino4f_bnred(dy.data_ptr(), Uf.data_ptr(), dx.data_ptr(), N, H, W, Cout, Cin, Cin, yP.data_ptr(),
                                           scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), part.data_ptr(), cap, s),
              "bnred")
        dbeta, dgamma = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
        check(lib.cvk_colsum_finalize(part.data_ptr(), P, Cin, dbeta.data_ptr(), dgamma.data_ptr(), s), "finalize")
        torch.cuda.synchronize()
        assert torch.equal(dx, plain)
        mask = (yP * scale + shift) > 0
        gm = torch.where(mask, dx, torch.zeros_like(dx)).double()
        xh = (yP.double() - mean.double()) * rstd.double()
        want_b, want_g = gm.sum(dim=(0, 1, 2)), (gm * xh).sum(dim=(0, 1, 2))
        sc = gm.abs().sum(dim=(0, 1, 2)).max().item()
        assert (dbeta.double() - want_b).abs().max().item() < 2e-5 * sc, cap
        assert (dgamma.double() - want_g).abs().max().item() < 2e-5 * max(sc, (gm * xh).abs().sum(dim=(0, 1, 2)).max().item()), cap


def test_fused_is_deterministic_and_tile_walk_independent():
    """Bitwise reproducible, and every image of a batch equals the same image run alone (the persistent tile walk and the
    batch size must not change any value: the K order per output is fixed)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 40, 72, 64, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(64, generator=g).cuda()
    y1, s1, _ = _fused_conv(x, w, b)
    y2, s2, _ = _fused_conv(x, w, b)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)
    for n in range(4):
        yn, _, _ = _fused_conv(x[n:n + 1].contiguous(), w, b)
        assert torch.equal(yn[0], y1[n])
    # a capped persistent grid (data-parallel runs leave CUs to RCCL) walks more tiles per workgroup: same values, same partials
    for cap in (1, 5, 37):
        yc, sc, _ = _fused_conv(x, w, b, max_wg=cap)
        assert torch.equal(yc, y1) and torch.equal(sc, s1), cap


# ------------------------------------------------------------------------- weight-grad through transform-domain planes (csrc/wgradp.hip)
def _planes_wgrad(x_nhwc, dy_nhwc):
    lib, check = _lib()
    s = torch.cuda.current_stream().cuda_stream
    N, H, W, Cin = x_nhwc.shape
    Cout = dy_nhwc.shape[3]
    dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
    wsb = lib.cvk_conv3x3_wgradp_workspace_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    check(lib.cvk_conv3x3_wgradp(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), None, dw.data_ptr(), N, H, W, Cin, Cin, Cout, Cout, ws.data_ptr(), wsb, s),
          "wgradp")
    torch.cuda.synchronize()
    return dw


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(1, 5, 7, 64, 64), (2, 9, 13, 64, 128), (2, 33, 50, 128, 64), (1, 16, 36, 128, 128),
                                             (3, 64, 96, 64, 64), (2, 3, 4, 256, 64), (2, 1, 40, 64, 64), (8, 90, 120, 64, 64)])
def test_planes_weight_grad_vs_fp64(N, H, W, Cin, Cout):
    """dW of conv2d(x, w, padding=1) for an upstream gradient dy — fp64 reference by autograd.  Ragged widths (W % 4, column
    groups not a multiple of 8), single-row images (every step starts a new strip), runs that cross strips and images."""
    g = torch.Generator().manual_seed(N + 7 * H + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, padding=1).backward(dy.double())
    ref = w.grad                                                                      # [Cout][Cin][3][3]
    dw = _planes_wgrad(x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda())
    got = dw.permute(0, 3, 1, 2).double().cpu()                                       # [Cout][3][3][Cin] -> OIHW
    assert torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 5e-6, rel                      # F(4,3) transform constants + an fp32 sum over N*H*W/4 products per element


def test_planes_weight_grad_is_deterministic():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 40, 72, 64, generator=g).cuda()
    dy = torch.randn(2, 40, 72, 128, generator=g).cuda()
    a, b = _planes_wgrad(x, dy), _planes_wgrad(x, dy)
    assert torch.equal(a, b)


def test_fused_f43_bitwise_equals_the_pre_packed_kernel():
    """commits eb2087d / 36c1d00 (input transform on channel pairs, packed FMAs, inline asm) claimed "bitwise the same
    values" in their messages.  tests/golden/wino4f_bits.npz holds bit-pattern checksums (1024 wrapping uint32 sums per tensor) of the
    output and statistics partials of the library built at bf80833 — the commit before them — on one ragged shape, forward (bias +
    statistics) and data-grad form; today's kernel must reproduce every bucket (tests/golden/make_wino4f_bits.py, raw C ABI)."""
    import ctypes
    import sys
    G = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, G)
    import make_wino4f_bits as M
    from pytorch_camvid_amd import _lib
    _lib.load()                                    # ABI stamp checked; the calls below go through a raw handle
    ref = dict(np.load(os.path.join(G, "wino4f_bits.npz")))
    got = M.run(ctypes.CDLL(_lib.LIB_PATH))
    for k in ("fwd_y", "fwd_stats", "dgrad_y"):
        bad = int((got[k] != ref[k]).sum())
        assert bad == 0, (k, bad, got[k + "_first"] if k + "_first" in got else None)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 13, 64, 64), (1, 16, 36, 64, 128), (2, 33, 50, 128, 64), (3, 64, 96, 64, 64),
                                             (1, 45, 60, 128, 136), (8, 90, 120, 64, 64), (1, 2, 3, 64, 64), (1, 5, 7, 32, 64)])
def test_forward_emitted_v_planes_feed_the_weight_grad(N, H, W, Cin, Cout):
    """Round 6: cvk_conv3x3_wino4f_vplanes = the fused forward launch that also leaves the weight-grad's transformed input as six
    SLICE-MAJOR planes (csrc/wino4f.hip FVpl).  (1) y, statistics and counts are BITWISE those of cvk_conv3x3_wino4f, for any grid cap;
    (2) the planes equal cvk_wgradp_planes_sm(x) — the stand-alone pass, same formulas with other rounding — to 1e-6 of the plane's scale, and
    every pad row is zero; (3) cvk_wgradp_gemm_sm on slice-major planes is bitwise cvk_wgradp_gemm on row-major planes of the same values, and
    the weight gradient from the forward-emitted planes matches fp64 (reference: backward of nn.Conv2d, models/unet.py:11, train.py:131)."""
    lib, check = _lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(N * 77 + H + Cin + Cout)
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    y0, st0, P = _fused_conv(x, w, b)
    wcl = w.permute(0, 2, 3, 1).contiguous()
    Uf = torch.empty(lib.cvk_wino4f_weight_floats(Cout, Cin), device="cuda")
    check(lib.cvk_wino4f_weight_transform(wcl.data_ptr(), Uf.data_ptr(), Cout, Cin, 0, s), "weight")
    rows = lib.cvk_wgradp_plane_rows(N, H, W)
    ref_pl = torch.full((6 * rows * Cin,), float("nan"), device="cuda")
    check(lib.cvk_wgradp_planes_sm(x.data_ptr(), Cin, ref_pl.data_ptr(), N, H, W, Cin, s), "planes_sm")
    for cap in (0, 3):
        y = torch.full((N, H, W, Cout), float("nan"), device="cuda")
        st = torch.zeros(2 * P * Cout + P, device="cuda")
        pl = torch.full((6 * rows * Cin,), float("nan"), device="cuda")
        check(lib.cvk_wgradp_zero_pads_sm(pl.data_ptr(), N, H, W, Cin, s), "zero_pads_sm")
        check(lib.cvk_conv3x3_wino4f_vplanes(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * Cout,
                                             pl.data_ptr(), N, H, W, Cin, Cout, Cout, cap, s), "vplanes")
        torch.cuda.synchronize()
        assert torch.equal(y, y0) and torch.equal(st, st0), cap
        assert torch.isfinite(pl).all()                                   # every element written: by the kernel or by the pad pass
        assert (pl - ref_pl).abs().max().item() <= 1e-6 * max(1.0, ref_pl.abs().max().item())
        assert torch.equal(pl == 0, ref_pl == 0) or (pl[ref_pl == 0].abs().max().item() == 0.0)        # pad rows are exact zeros
    if Cin % 64 or Cout % 64:
        return
    # the weight gradient: E planes from dy (stand-alone pass), V planes from the forward launch
    dy = torch.randn(N, H, W, Cout, generator=g).cuda()
    E6 = torch.empty(6 * rows * Cout, device="cuda")
    check(lib.cvk_wgradp_planes(dy.data_ptr(), Cout, E6.data_ptr(), N, H, W, Cout, 1, s), "planes(dy)")
    wsb = lib.cvk_wgradp_gemm_workspace_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dw_sm = torch.empty(Cout, 3, 3, Cin, device="cuda")
    check(lib.cvk_wgradp_gemm_sm(E6.data_ptr(), pl.data_ptr(), dw_sm.data_ptr(), N, H, W, Cin, Cin, Cout, ws.data_ptr(), wsb, s), "gemm_sm")
    # same plane VALUES in the row-major layout -> the row-major GEMM must give the same bits
    rm = pl.view(6, Cin // 16, rows, 16).permute(0, 2, 1, 3).contiguous().view(-1)
    dw_rm = torch.empty(Cout, 3, 3, Cin, device="cuda")
    check(lib.cvk_wgradp_gemm(E6.data_ptr(), rm.data_ptr(), dw_rm.data_ptr(), N, H, W, Cin, Cin, Cout, ws.data_ptr(), wsb, s), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(dw_sm, dw_rm)
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
    want = wr.grad.permute(0, 2, 3, 1)
    rel = ((dw_sm.double().cpu() - want).norm() / want.norm()).item()
    assert rel < 1e-5, rel


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 12, 64, 64), (2, 33, 52, 128, 64), (1, 16, 36, 64, 128), (3, 64, 96, 64, 64), (1, 2, 4, 64, 64),
                                             (2, 9, 13, 64, 64)])
def test_plane_gemm_reads_the_identity_planes_from_dy(N, H, W, Cin, Cout):
    """Round 6: E0 and E5 of E = A dy are columns 4 xt and 4 xt + 3 of dy.  (1) cvk_bn_bwd_dx_e4p writes dy and the four planes E1..E4 bitwise as
    cvk_bn_bwd_dx_e6 writes dy and its planes 1..4; (2) cvk_wgradp_gemm_sm_dy (four planes + dy with a zeroed slack) gives bitwise the weight
    gradient of cvk_wgradp_gemm_sm on all six planes — also when the slack holds the only finite values beside garbage-free zeros, i.e. the
    pad column groups of the last row really read it (W = 12, 52, 36, 4: W/4 is not a multiple of 8).  W % 4 != 0 is refused: a ragged last column
    group has E5 = 0 where dy's next row begins."""
    from pytorch_camvid_amd._lib import View
    lib, check = _lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(N + 3 * H + 5 * W + Cin)
    M = N * H * W
    dout = torch.randn(N, H, W, Cout, generator=g).cuda()
    y = torch.randn(M, Cout, generator=g).cuda()
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
    mean, rstd = y.mean(0), (y.var(0, unbiased=False) + 1e-5).rsqrt()
    scale = gamma * rstd
    shift = beta - mean * scale
    dgam, dbet = torch.randn(Cout, generator=g).cuda(), torch.randn(Cout, generator=g).cuda()
    rows = lib.cvk_wgradp_plane_rows(N, H, W)
    PB = lib.cvk_bn_bwd_e_blocks(N, H, W)
    view = View(dout.data_ptr(), H * W * Cout, W * Cout, Cout)
    slack = lib.cvk_wgradp_dy_slack(W) * Cout
    res = {}
    for four in (False, True):
        dyb = torch.full((M * Cout + slack,), float("nan"), device="cuda")
        dyb[M * Cout:].zero_()
        E = torch.full(((4 if four else 6) * rows * Cout,), float("nan"), device="cuda")
        part = torch.zeros(PB * Cout, device="cuda")
        check((lib.cvk_wgradp_zero_pads4 if four else lib.cvk_wgradp_zero_pads)(E.data_ptr(), N, H, W, Cout, s), "zero pads")
        check((lib.cvk_bn_bwd_dx_e4p if four else lib.cvk_bn_bwd_dx_e6)(
            view, y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
            dyb.data_ptr(), Cout, E.data_ptr(), part.data_ptr(), N, H, W, Cout, 1, s), "bn_bwd_dx_e")
        torch.cuda.synchronize()
        assert torch.isfinite(dyb).all() and torch.isfinite(E).all()
        res[four] = (dyb, E, part)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])
    assert torch.equal(res[True][1], res[False][1][rows * Cout:5 * rows * Cout])
    # the weight-grad: V planes from x (stand-alone slice-major pass)
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    V = torch.empty(6 * rows * Cin, device="cuda")
    check(lib.cvk_wgradp_planes_sm(x.data_ptr(), Cin, V.data_ptr(), N, H, W, Cin, s), "planes_sm")
    wsb = lib.cvk_wgradp_gemm_workspace_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dw6, dw4 = torch.empty(Cout, 9 * Cin, device="cuda"), torch.empty(Cout, 9 * Cin, device="cuda")
    check(lib.cvk_wgradp_gemm_sm(res[False][1].data_ptr(), V.data_ptr(), dw6.data_ptr(), N, H, W, Cin, Cin, Cout, ws.data_ptr(), wsb, s), "gemm_sm")
    if W % 4:
        rc = lib.cvk_wgradp_gemm_sm_dy(res[True][1].data_ptr(), res[True][0].data_ptr(), V.data_ptr(), dw4.data_ptr(), N, H, W, Cin, Cin, Cout,
                                       ws.data_ptr(), wsb, s)
        assert rc == -1 and b"multiple of 4" in lib.cvk_last_error_string()
        return
    check(lib.cvk_wgradp_gemm_sm_dy(res[True][1].data_ptr(), res[True][0].data_ptr(), V.data_ptr(), dw4.data_ptr(), N, H, W, Cin, Cin, Cout,
                                    ws.data_ptr(), wsb, s), "gemm_sm_dy")
    torch.cuda.synchronize()
    assert torch.isfinite(dw4).all() and torch.equal(dw4, dw6)
