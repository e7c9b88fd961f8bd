"""Raw C-ABI parity of the 2-D Winograd pipelines (csrc/wino2d.hip) — F(4x4,3x3) `cvk_w2d_*` and F(6x6,3x3) `cvk_w6_*` — pass by
pass against the fp64 operator they replace: nn.Conv2d(cin, cout, 3, padding=1) of /root/reference/models/unet.py:11 (forward +
the BatchNorm batch statistics of unet.py:12), its data-grad and its weight-grad (backward of train.py:131).
Tolerances: fp32 rounding of the transforms, measured with a numpy restatement at 64 / 256 input channels: F(4x4) 1.4e-6 / 2.7e-6,
F(6x6) 2.7e-6 / 5.1e-6 relative L2 (it grows ~sqrt(Cin)); bounds below are 3x those."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from pytorch_camvid_amd import _lib
    return _lib.load(), _lib.check


def _s():
    return torch.cuda.current_stream().cuda_stream


def fam(lib, mt):
    pre = "cvk_w2d_" if mt == 4 else "cvk_w6_"
    return lambda name: getattr(lib, pre + name)


TOL = {4: 1.0e-5, 6: 1.8e-5}

CASES = [  # N, H, W, Cin, Cout: ragged tiles (H, W not multiples of 4 / 6), one tile, K-split tails, >128 output channels
    (1, 5, 7, 32, 64),
    (2, 12, 18, 64, 64),
    (1, 22, 30, 256, 256),
    (2, 45, 60, 128, 192),
    (1, 3, 4, 512, 128),
    (8, 22, 30, 512, 256),
    (2, 90, 120, 64, 128),
]


def _forward(mt, x_nhwc, w_oihw, bias, stats=True, dgrad=False):
    lib, check = _lib()
    f = fam(lib, mt)
    N, H, W, Ck = x_nhwc.shape
    wcl = w_oihw.permute(0, 2, 3, 1).contiguous()
    Cn = w_oihw.shape[1] if dgrad else w_oihw.shape[0]
    nx = (mt + 2) ** 2
    U = torch.empty(nx, Cn, Ck, device="cuda")
    if dgrad:
        check(f("weight_transform_dgrad")(wcl.data_ptr(), U.data_ptr(), w_oihw.shape[0], w_oihw.shape[1], _s()), "weight(dgrad)")
    else:
        check(f("weight_transform")(wcl.data_ptr(), U.data_ptr(), Cn, Ck, _s()), "weight")
    T = f("tiles")(N, H, W)
    Tp = lib.cvk_w2d_tpad(T)
    ks = f("ksplit")(T, Ck, Cn)
    V = torch.full((nx * Tp * Ck + 128,), float("nan"), device="cuda")
    Mo = torch.full((ks * nx * T * Cn,), float("nan"), device="cuda")
    wsb = lib.cvk_conv3x3_w2d_workspace_bytes(N, H, W, Ck, Cn) if mt == 4 else lib.cvk_conv3x3_w6_workspace_bytes(N, H, W, Ck, Cn)
    assert wsb == 4 * (nx * Tp * Ck + 128 + ks * nx * T * Cn)
    y = torch.full((N, H, W, Cn), float("nan"), device="cuda")
    P = f("stat_partials")(N, H, W)
    st = torch.full((2 * P * Cn + P,), float("nan"), device="cuda") if stats else None
    check(f("input_transform")(x_nhwc.data_ptr(), V.data_ptr(), N, H, W, Ck, _s()), "input")
    check(f("gemm")(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), T, Ck, Cn, _s()), "gemm")
    check(f("output")(Mo.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), st.data_ptr() if stats else None,
                      st.data_ptr() + 8 * P * Cn if stats else None, N, H, W, Ck, Cn, Cn, _s()), "output")
    torch.cuda.synchronize()
    return y, st, P, V


@pytest.mark.parametrize("mt", [4, 6])
@pytest.mark.parametrize("N,H,W,Cin,Cout", CASES)
def test_forward_and_statistics_vs_fp64(mt, N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cin + mt)
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g))                # activations of the net are post-ReLU
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y, st, P, _ = _forward(mt, x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), b.cuda())
    got = y.permute(0, 3, 1, 2).double().cpu()
    assert torch.isfinite(got).all()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < TOL[mt], rel
    M = N * H * W
    sums = st[:P * Cout].view(P, Cout).double().cpu()
    m2 = st[P * Cout:2 * P * Cout].view(P, Cout).double().cpu()
    cnt = st[2 * P * Cout:].double().cpu()
    assert int(cnt.sum().item()) == M
    mean = sums.sum(0) / M
    var = (m2.sum(0) + (cnt[:, None] * (sums / cnt[:, None] - mean) ** 2).sum(0)) / M
    rmean, rvar = ref.mean(dim=(0, 2, 3)), ref.var(dim=(0, 2, 3), unbiased=False)
    assert (mean - rmean).abs().max().item() < 1e-5 * max(1.0, rmean.abs().max().item())
    assert ((var - rvar).abs() <= 1e-4 * rvar + 1e-6 * (ref ** 2).mean(dim=(0, 2, 3))).all()


@pytest.mark.parametrize("mt", [4, 6])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 12, 18, 64, 64), (1, 22, 30, 256, 128), (2, 45, 60, 128, 192)])
def test_data_grad_vs_fp64(mt, N, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(7 + Cin + Cout + mt)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    dy = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    dx, _, _, _ = _forward(mt, dy.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), None, stats=False, dgrad=True)
    got = dx.permute(0, 3, 1, 2).double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < TOL[mt], rel


@pytest.mark.parametrize("mt", [4, 6])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 12, 18, 64, 64), (1, 22, 30, 256, 128), (2, 45, 60, 128, 192), (8, 22, 30, 512, 256)])
def test_weight_grad_vs_fp64(mt, N, H, W, Cin, Cout):
    lib, check = _lib()
    f = fam(lib, mt)
    g = torch.Generator().manual_seed(11 + Cin + Cout + mt)
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g))
    dy = torch.randn(N, Cout, H, W, generator=g)
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wref, padding=1).backward(dy.double())
    ref = wref.grad.permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
    nx = (mt + 2) ** 2
    T = f("tiles")(N, H, W)
    Tp = lib.cvk_w2d_tpad(T)
    fs = f("wgrad_ksplit")(T, Cin, Cout)
    V = torch.full((nx * Tp * Cin + 128,), float("nan"), device="cuda")
    E = torch.full((nx * Tp * Cout + 128,), float("nan"), device="cuda")
    Pp = torch.full((fs * nx * Cout * Cin,), float("nan"), device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
        check(f("input_transform")(xn.data_ptr(), V.data_ptr(), N, H, W, Cin, _s()), "input")
        check(f("dy_transform")(dyn.data_ptr(), Cout, E.data_ptr(), N, H, W, Cout, _s()), "dy")
        check(f("gemm_tn")(E.data_ptr(), V.data_ptr(), Pp.data_ptr(), T, Cin, Cout, _s()), "gemm_tn")
        check(f("wgrad_output")(Pp.data_ptr(), dw.data_ptr(), T, Cin, Cin, Cout, _s()), "wgrad_out")
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    got = outs[0].double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2 * TOL[mt], rel


@pytest.mark.parametrize("mt", [4, 6])
@pytest.mark.parametrize("N,H,W,C", [(2, 12, 18, 64), (1, 22, 30, 256), (2, 45, 60, 128), (1, 5, 7, 96)])
def test_both_transforms_of_dy_in_one_launch(mt, N, H, W, C):
    """cvk_w*_dy_transform_both == (cvk_w*_input_transform(dy), cvk_w*_dy_transform(dy)) bit for bit: the backward pass of a block whose
    data-grad and weight-grad both run the 2-D path (backward of nn.Conv2d, train.py:131) transforms dy once."""
    lib, check = _lib()
    f = fam(lib, mt)
    g = torch.Generator().manual_seed(mt + H + W + C)
    dy = torch.randn(N, H, W, C, generator=g).cuda()
    nx = (mt + 2) ** 2
    T = f("tiles")(N, H, W); Tp = lib.cvk_w2d_tpad(T)
    n = nx * Tp * C
    V0 = torch.full((n + 128,), float("nan"), device="cuda"); E0 = torch.full((n + 128,), float("nan"), device="cuda")
    V1 = torch.full((n + 128,), float("nan"), device="cuda"); E1 = torch.full((n + 128,), float("nan"), device="cuda")
    check(f("input_transform")(dy.data_ptr(), V0.data_ptr(), N, H, W, C, _s()), "input")
    check(f("dy_transform")(dy.data_ptr(), C, E0.data_ptr(), N, H, W, C, _s()), "dy")
    check(f("dy_transform_both")(dy.data_ptr(), C, V1.data_ptr(), E1.data_ptr(), N, H, W, C, _s()), "both")
    torch.cuda.synchronize()
    assert torch.equal(V1[:n], V0[:n]) and torch.equal(E1[:n], E0[:n])


@pytest.mark.parametrize("T,Cin,Cout", [(160, 64, 256), (2400, 128, 128), (20, 512, 384), (300, 32, 192)])
def test_gemm_store_paths_keep_to_their_plane(T, Cin, Cout):
    """Round 6: full-width tiles of k_w2d_gemm leave as raw buffer stores whose row bound (row < T) is the descriptor's range check.  Raw C-ABI
    call with ragged T: every plane equals the fp64 product, rows T.. of the last 128-row tile never reach the next plane (a guard band behind
    the product planes stays untouched), and two launches agree bitwise.  Cout = 192: the checked form for the ragged column tile."""
    lib, check = _lib()
    nx = 64
    g = torch.Generator().manual_seed(T + Cin + Cout)
    Tp = lib.cvk_w2d_tpad(T)
    ks = lib.cvk_w6_ksplit(T, Cin, Cout)
    V = torch.zeros(nx, Tp, Cin)
    V[:, :T] = torch.randn(nx, T, Cin, generator=g)
    U = torch.randn(nx, Cout, Cin, generator=g) * 0.1
    Vd = torch.cat([V.reshape(-1), torch.zeros(128)]).cuda()
    Ud = U.cuda()
    guard = 4096
    out = []
    for _ in range(2):
        Mo = torch.zeros(ks * nx * T * Cout + guard, device="cuda")       # (parts 1.. exist for the tail tiles only: zero elsewhere)
        Mo[-guard:] = -7.0
        check(lib.cvk_w6_gemm(Vd.data_ptr(), Ud.data_ptr(), Mo.data_ptr(), T, Cin, Cout, _s()), "gemm")
        torch.cuda.synchronize()
        assert (Mo[-guard:] == -7.0).all()
        out.append(Mo[:-guard].reshape(ks, nx, T, Cout).cpu())
    assert torch.equal(out[0], out[1])
    got = out[0].double().sum(0)                         # the K-split parts of the tail tiles add up in the output transform
    ref = torch.einsum("xtk,xck->xtc", V[:, :T].double(), U.double())
    per_plane = (got - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert per_plane.max().item() < 2e-6, per_plane.max().item()
