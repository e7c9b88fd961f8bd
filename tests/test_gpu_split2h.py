"""OPT-IN split-operand path, format 2 (csrc/split_fmt.h, csrc/split3.hip; runner.w2d_split = 2 — never the default): the GEMM stage of the
2-D Winograd path with TWO fp16 terms per fp32 operand, scaled per transform index by an exact power of two derived from the largest
magnitude of the tensor the transform reads, three cross-products per fp32 product on v_mfma_f32_16x16x32_f16 with fp32 accumulation.
Through the C ABI against fp64: (1) the maximum pass is exact ("amax blocks", csrc/cvk_common.h); (2) the planes hold 2^e x = h1 + h2 to 22 bits with no value outside
fp16's range whatever the magnitude of the data (1e-30 .. 1e+30) and the exponent is the one cvk_split_scale_exponent reports; (3) the
batched GEMM and (4) the weight-grad GEMM against the exact-fp32 GEMMs of the default path; (5) a whole conv layer — transform kernels that
write the planes themselves, split GEMM, plain output pass — against an fp64 convolution of the reference operator (nn.Conv2d(3x3,
padding=1), models/unet.py:11), at ordinary, tiny, huge and heavy-tailed magnitudes; (6) the headline network against the
REFERENCE-generated fixtures."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from pytorch_camvid_amd import _lib
    return _lib.load(), _lib.check


def _amax(lib, check, t, s, C=None):
    """the cvk_absmax_f32 word of a dense tensor"""
    C = t.shape[-1] if C is None else C
    a = _block(lib, t.device)
    check(lib.cvk_absmax_f32(t.data_ptr(), t.numel() // C, C, C, a.data_ptr(), s), "cvk_absmax_f32")
    return a


def _block(lib, dev):
    """a zeroed amax block (csrc/cvk_common.h)"""
    return torch.zeros(lib.cvk_amax_block_words(), device=dev, dtype=torch.int32)


def _val(a):
    """the value of an amax block as a float: the maximum over its slots (bit patterns of non-negative floats order like integers)"""
    return a[::32].max().view(torch.float32).item()


def _unswizzle(S, NX, C, Rp):
    """split planes [xi][C/32][2][Rp][32 swizzled] -> [2][NX][Rp][C] (float64)"""
    S = S.view(NX, C // 32, 2, Rp, 4, 8).double().cpu()
    r = torch.arange(Rp)
    pos = (torch.arange(4)[None, :] ^ (((r >> 2) & 1) << 1)[:, None])
    un = torch.gather(S, 4, pos[None, None, None, :, :, None].expand(NX, C // 32, 2, Rp, 4, 8))
    return un.reshape(NX, C // 32, 2, Rp, 32).permute(2, 0, 3, 1, 4).reshape(2, NX, Rp, C)


def test_absmax_is_exact_and_combines_by_atomic_max():
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1000, 52, generator=g) * torch.exp(4 * torch.randn(1000, 52, generator=g))).to(dev)
    a = _amax(lib, check, x, s)
    assert _val(a) == x.abs().max().item()
    # a strided view: only the first 20 of 52 columns count
    a2 = _block(lib, dev)
    check(lib.cvk_absmax_f32(x.data_ptr(), 1000, 20, 52, a2.data_ptr(), s), "absmax strided")
    assert _val(a2) == x[:, :20].abs().max().item()
    # a second tensor into the same word: the larger value stays
    y = torch.full((8, 4), -3e30, device=dev)
    check(lib.cvk_absmax_f32(y.data_ptr(), 8, 4, 4, a2.data_ptr(), s), "absmax second")
    assert _val(a2) == torch.tensor(3e30).item()
    z = torch.zeros(64, 4, device=dev)
    a3 = _amax(lib, check, z, s)
    assert _val(a3) == 0 and int(a3.abs().max()) == 0
    host = a2.cpu().numpy().astype(np.uint32)
    import ctypes
    assert lib.cvk_amax_block_value(host.ctypes.data_as(ctypes.c_void_p), host.size) == int(a2[::32].max().item())


@pytest.mark.parametrize("tile,kind", [(6, 0), (6, 1), (6, 2), (4, 0)])
@pytest.mark.parametrize("mag", [1.0, 1e-30, 1e30])
def test_planes_hold_the_scaled_value_inside_fp16_range(tile, kind, mag):
    """cvk_split_planes(fmt 2): h1 + h2 = 2^e * value to 22 bits for every value within 2^17 of its plane's bound; |h1| < 2^15 always (the
    planes here are filled with values up to the worst case a transform of `kind` can produce from a tensor with this maximum)."""
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    NX, R, C = (tile + 2) ** 2, 37, 64
    g = torch.Generator().manual_seed(tile + kind)
    src = torch.tensor([mag, -0.25 * mag, 0.0, 0.0], device=dev)            # "the tensor the transform read": its maximum is mag
    am = _amax(lib, check, src.view(1, 4), s)
    word = int(am[::32].max().item()) & 0xFFFFFFFF
    e = torch.tensor([lib.cvk_split_scale_exponent(tile, kind, xi, word) for xi in range(NX)], dtype=torch.float64)
    bound = 2.0 ** (15 - e)                                                   # what |value| may reach in plane xi
    P = (torch.rand(NX, R, C, generator=g, dtype=torch.float64) * 2 - 1) * torch.exp(-6 * torch.rand(NX, R, C, generator=g, dtype=torch.float64))
    P[:, 0, 0] = 0.999                                                        # and one value right at the bound
    P = (P * bound[:, None, None]).float().to(dev)
    Rp = lib.cvk_split3_rows_pad(R, 256)
    S = torch.full((NX * (C // 32) * 2 * Rp * 32,), float("nan"), device=dev, dtype=torch.float16)
    check(lib.cvk_split_planes(2, tile, kind, P.data_ptr(), S.data_ptr(), am.data_ptr(), NX, R, Rp, C, s), "cvk_split_planes")
    assert torch.isfinite(S).all()
    un = _unswizzle(S, NX, C, Rp)
    assert torch.all(un[:, :, R:] == 0)
    assert un[0].abs().max().item() < 2.0 ** 15
    tot = (un[0] + un[1])[:, :R] * (2.0 ** -e)[:, None, None]
    ref = P.double().cpu()
    big = ref.abs() > bound[:, None, None] * 2.0 ** -17
    rel = ((tot - ref).abs() / ref.abs().clamp_min(1e-300))[big].max().item()
    assert rel < 2.0 ** -21, rel
    absr = ((tot - ref).abs() / bound[:, None, None])[~big].max().item()     # small values: absolute error far below the plane's bound
    assert absr < 2.0 ** -38, absr


@pytest.mark.parametrize("T,Cin,Cout", [(300, 64, 128), (2400, 256, 256), (530, 512, 128)])
@pytest.mark.parametrize("heavy", [False, True])
def test_split_gemm_vs_the_fp32_gemm(T, Cin, Cout, heavy):
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    tile, NX = 6, 64
    g = torch.Generator().manual_seed(T)
    Tp32 = lib.cvk_w2d_tpad(T)
    V = torch.zeros(NX, Tp32, Cin)
    V[:, :T] = torch.randn(NX, T, Cin, generator=g).clamp_min(-0.5)
    U = torch.randn(NX, Cout, Cin, generator=g) / (Cin ** 0.5)
    if heavy:                                                                 # a few values 1000x the rest: most of the plane far below its maximum
        V[:, :T] *= torch.exp(2.5 * torch.randn(NX, T, Cin, generator=g))
    Vd, Ud = V.to(dev), U.to(dev)
    ref = torch.einsum("xtc,xoc->xto", Vd[:, :T].double(), Ud.double())
    f = lib.cvk_w6_ksplit(T, Cin, Cout)
    Mo = torch.zeros(f, NX, T, Cout, device=dev)
    Vp = torch.cat([Vd.reshape(-1), torch.zeros(128, device=dev)])
    check(lib.cvk_w6_gemm(Vp.data_ptr(), Ud.data_ptr(), Mo.data_ptr(), T, Cin, Cout, s), "cvk_w6_gemm")
    e32 = ((Mo.sum(0).double() - ref).norm() / ref.norm()).item()
    # the planes are scaled as planes of kind B / G of tensors whose maxima make every plane fit: amax = plane maximum (row sums >= 1 only
    # widen the margin; G rows below 1 narrow it by at most 2^6 in 2-D: feed max / 2^-6)
    amv = _amax(lib, check, Vd, s)
    amu = _amax(lib, check, (Ud * 64.0).contiguous(), s)
    Tp = lib.cvk_split3_rows_pad(T, 256); Cp = lib.cvk_split3_rows_pad(Cout, 128)
    V2 = torch.empty(NX * (Cin // 32) * 2 * Tp * 32, device=dev, dtype=torch.float16)
    U2 = torch.empty(NX * (Cin // 32) * 2 * Cp * 32, device=dev, dtype=torch.float16)
    check(lib.cvk_split_planes(2, tile, 0, Vd[:, :T].contiguous().data_ptr(), V2.data_ptr(), amv.data_ptr(), NX, T, Tp, Cin, s), "split V")
    check(lib.cvk_split_planes(2, tile, 1, Ud.data_ptr(), U2.data_ptr(), amu.data_ptr(), NX, Cout, Cp, Cin, s), "split U")
    assert torch.isfinite(V2).all() and torch.isfinite(U2).all()
    M2 = torch.full((NX, T, Cout), float("nan"), device=dev)
    check(lib.cvk_w2d_gemm_split(2, tile, V2.data_ptr(), U2.data_ptr(), M2.data_ptr(), amv.data_ptr(), amu.data_ptr(), NX, T, Tp, Cin, Cout, Cp, s),
          "cvk_w2d_gemm_split")
    assert torch.isfinite(M2).all()
    e2 = ((M2.double() - ref).norm() / ref.norm()).item()
    print(f"T={T} {Cin}->{Cout} heavy={heavy}: relative L2 vs fp64: exact-fp32 GEMM {e32:.2e}, 2-term fp16 split GEMM {e2:.2e}")
    assert e2 < 6e-7 and e2 <= 1.5 * e32 + 1e-7, (e2, e32)
    M2b = torch.empty_like(M2)
    check(lib.cvk_w2d_gemm_split(2, tile, V2.data_ptr(), U2.data_ptr(), M2b.data_ptr(), amv.data_ptr(), amu.data_ptr(), NX, T, Tp, Cin, Cout, Cp, s),
          "cvk_w2d_gemm_split")
    assert torch.equal(M2, M2b)                                               # bitwise reproducible


def _layer(lib, check, tile, x, w, dev, s):
    """conv3x3 of x [N,H,W,Cin] with w [Cout][3][3][Cin] through the fmt-2 kernels and through the default fp32 kernels -> (y2, y32)"""
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    NX = 64 if tile == 6 else 36
    fam = "cvk_w6_" if tile == 6 else "cvk_w2d_"
    fn = lambda name: getattr(lib, fam + name)
    T = fn("tiles")(N, H, W); Tp32 = lib.cvk_w2d_tpad(T)
    bias = torch.zeros(Cout, device=dev)
    V = torch.zeros(NX * Tp32 * Cin + 128, device=dev); U = torch.empty(NX * Cout * Cin, device=dev)
    check(fn("input_transform")(x.data_ptr(), V.data_ptr(), N, H, W, Cin, s), "input")
    check(fn("weight_transform")(w.data_ptr(), U.data_ptr(), Cout, Cin, s), "weight")
    f = fn("ksplit")(T, Cin, Cout)
    Mo = torch.zeros(f * NX * T * Cout, device=dev)
    check(fn("gemm")(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), T, Cin, Cout, s), "gemm")
    y32 = torch.empty(N, H, W, Cout, device=dev)
    check(fn("output")(Mo.data_ptr(), bias.data_ptr(), y32.data_ptr(), None, None, N, H, W, Cin, Cout, Cout, s), "output")
    amx, amw = _amax(lib, check, x, s), _amax(lib, check, w, s)
    Tp = lib.cvk_split3_rows_pad(T, 256); Cp = lib.cvk_split3_rows_pad(Cout, 128)
    V2 = torch.full((NX * (Cin // 32) * 2 * Tp * 32,), float("nan"), device=dev, dtype=torch.float16)
    U2 = torch.zeros(NX * (Cin // 32) * 2 * Cp * 32, device=dev, dtype=torch.float16)
    check(lib.cvk_w2d_input_transform_split(2, tile, x.data_ptr(), V2.data_ptr(), amx.data_ptr(), N, H, W, Cin, s), "input split")
    check(lib.cvk_w2d_weight_transform_split(2, tile, w.data_ptr(), U2.data_ptr(), amw.data_ptr(), Cout, Cin, 0, s), "weight split")
    assert torch.isfinite(V2).all() and torch.isfinite(U2).all()
    M2 = torch.full((NX * T * Cout,), float("nan"), device=dev)
    check(lib.cvk_w2d_gemm_split(2, tile, V2.data_ptr(), U2.data_ptr(), M2.data_ptr(), amx.data_ptr(), amw.data_ptr(), NX, T, Tp, Cin, Cout, Cp, s), "gemm split")
    y2 = torch.empty(N, H, W, Cout, device=dev)
    check(lib.cvk_w2d_output_plain(tile, M2.data_ptr(), bias.data_ptr(), y2.data_ptr(), None, None, N, H, W, Cout, Cout, s), "output plain")
    return y2, y32


@pytest.mark.parametrize("tile", [6, 4])
@pytest.mark.parametrize("case", ["relu", "tiny", "huge", "heavy", "sparse"])
def test_whole_layer_vs_fp64_conv(tile, case):
    """256 -> 256 channels at 2 x 45 x 60 through the fmt-2 kernels (transforms writing the planes, split GEMM, plain output pass) against the fp64
    convolution, the default fp32 kernels beside it.  The scale follows the data: activations of magnitude 1e-25 or 1e+25, a heavy-tailed
    tensor (a few values 1e4 x the rest) and a tensor that is zero except for a few pixels all stay within 1.35x of the fp32 path's error."""
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    N, H, W, Cin, Cout = 2, 45, 60, 256, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).clamp_min(0)
    if case == "tiny": x = x * 1e-25
    if case == "huge": x = x * 1e25
    if case == "heavy": x = x * torch.exp(3 * torch.randn(N, H, W, Cin, generator=g))
    if case == "sparse": x = x * (torch.rand(N, H, W, 1, generator=g) < 0.01)
    x = x.to(dev)
    w = ((torch.rand(Cout, 3, 3, Cin, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    y2, y32 = _layer(lib, check, tile, x, w, dev, s)
    assert torch.isfinite(y2).all()
    e2 = ((y2.double() - ref).norm() / ref.norm()).item(); e32 = ((y32.double() - ref).norm() / ref.norm()).item()
    print(f"layer 256->256 @2x45x60 F({tile}x{tile},3x3) {case}: relative L2 vs fp64: fp32 kernels {e32:.2e}, 2-term fp16 split {e2:.2e}")
    assert e2 <= 1.35 * e32 and e2 < (8e-6 if tile == 6 else 4e-6), (e2, e32)


@pytest.mark.parametrize("tile", [6, 4])
def test_backward_transforms_and_gemms_vs_fp64(tile):
    """dy -> V' and E in one launch (fmt 2), the data-grad through the split GEMM with the rotated filter, and the weight-grad GEMM E^T V with
    its final pass — against fp64 autograd of the reference operator, the default fp32 kernels' errors beside them."""
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    N, H, W, Cin, Cout = 2, 45, 60, 256, 256
    NX = 64 if tile == 6 else 36
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, H, W, Cin, generator=g).clamp_min(0).to(dev)
    w = ((torch.rand(Cout, 3, 3, Cin, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev)
    dy = (torch.randn(N, H, W, Cout, generator=g) * torch.exp(1.5 * torch.randn(N, H, W, Cout, generator=g)) * 1e-6).to(dev)   # gradients: small, long-tailed
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(True); wr = w.permute(0, 3, 1, 2).double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, padding=1).backward(dy.permute(0, 3, 1, 2).double())
    dx_ref = xr.grad.permute(0, 2, 3, 1); dw_ref = wr.grad.permute(0, 2, 3, 1)
    fam = "cvk_w6_" if tile == 6 else "cvk_w2d_"
    T = getattr(lib, fam + "tiles")(N, H, W)
    Tp = lib.cvk_split3_rows_pad(T, 256)
    amx, amw, amd = _amax(lib, check, x, s), _amax(lib, check, w, s), _amax(lib, check, dy, s)
    nel = lambda C: NX * (C // 32) * 2 * Tp * 32
    V2 = torch.empty(nel(Cin), device=dev, dtype=torch.float16)
    check(lib.cvk_w2d_input_transform_split(2, tile, x.data_ptr(), V2.data_ptr(), amx.data_ptr(), N, H, W, Cin, s), "input split")
    Vp2 = torch.full((nel(Cout),), float("nan"), device=dev, dtype=torch.float16); E2 = torch.full((nel(Cout),), float("nan"), device=dev, dtype=torch.float16)
    check(lib.cvk_w2d_dy_transform_both_split(2, tile, dy.data_ptr(), Cout, Vp2.data_ptr(), E2.data_ptr(), 1, amd.data_ptr(), N, H, W, Cout, s), "dy both split")
    assert torch.isfinite(Vp2).all() and torch.isfinite(E2).all()
    # data-grad: dX = conv(dy, rotated filter) — U' rows = Cin, depth = Cout
    Cp = lib.cvk_split3_rows_pad(Cin, 128)
    Ud = torch.zeros(NX * (Cout // 32) * 2 * Cp * 32, device=dev, dtype=torch.float16)
    check(lib.cvk_w2d_weight_transform_split(2, tile, w.data_ptr(), Ud.data_ptr(), amw.data_ptr(), Cout, Cin, 1, s), "weight split dgrad")
    M2 = torch.full((NX * T * Cin,), float("nan"), device=dev)
    check(lib.cvk_w2d_gemm_split(2, tile, Vp2.data_ptr(), Ud.data_ptr(), M2.data_ptr(), amd.data_ptr(), amw.data_ptr(), NX, T, Tp, Cout, Cin, Cp, s), "dgrad gemm")
    dx = torch.empty(N, H, W, Cin, device=dev)
    check(lib.cvk_w2d_output_plain(tile, M2.data_ptr(), None, dx.data_ptr(), None, None, N, H, W, Cin, Cin, s), "output plain")
    e_dx = ((dx.double() - dx_ref).norm() / dx_ref.norm()).item()
    # weight-grad
    f = lib.cvk_w2d_gemm_tn_split3_ksplit(NX, Tp, Cin, Cout)
    P = torch.full((f * NX * Cout * Cin,), float("nan"), device=dev)
    check(lib.cvk_w2d_gemm_tn_split(2, tile, E2.data_ptr(), V2.data_ptr(), P.data_ptr(), amd.data_ptr(), amx.data_ptr(), NX, Tp, Cin, Cout, s), "gemm_tn split")
    assert torch.isfinite(P).all()
    dw = torch.empty(Cout, 3, 3, Cin, device=dev)
    check(lib.cvk_w2d_wgrad_output_f(tile, P.data_ptr(), dw.data_ptr(), Cin, Cin, Cout, f, s), "wgrad out")
    e_dw = ((dw.double() - dw_ref).norm() / dw_ref.norm()).item()
    # the default fp32 kernels on the same data
    Tp32 = lib.cvk_w2d_tpad(T)
    V = torch.zeros(NX * Tp32 * Cin + 128, device=dev); Vp = torch.zeros(NX * Tp32 * Cout + 128, device=dev); E = torch.zeros(NX * Tp32 * Cout + 128, device=dev)
    check(getattr(lib, fam + "input_transform")(x.data_ptr(), V.data_ptr(), N, H, W, Cin, s), "input")
    check(getattr(lib, fam + "dy_transform_both")(dy.data_ptr(), Cout, Vp.data_ptr(), E.data_ptr(), N, H, W, Cout, s), "dy both")
    f32 = getattr(lib, fam + "wgrad_ksplit")(T, Cin, Cout)
    P32 = torch.zeros(f32 * NX * Cout * Cin, device=dev)
    check(getattr(lib, fam + "gemm_tn")(E.data_ptr(), V.data_ptr(), P32.data_ptr(), T, Cin, Cout, s), "gemm_tn")
    dw32 = torch.empty_like(dw)
    check(getattr(lib, fam + "wgrad_output")(P32.data_ptr(), dw32.data_ptr(), T, Cin, Cin, Cout, s), "wgrad out")
    e_dw32 = ((dw32.double() - dw_ref).norm() / dw_ref.norm()).item()
    print(f"F({tile}x{tile}) backward 256->256 @2x45x60: dX rel L2 {e_dx:.2e}; dW rel L2 {e_dw:.2e} (fp32 kernels {e_dw32:.2e})")
    assert e_dx < (8e-6 if tile == 6 else 4e-6), e_dx
    assert e_dw <= 1.35 * e_dw32 + 1e-7 and e_dw < 1e-5, (e_dw, e_dw32)


def test_unet_headline_step_in_the_network():
    """runner.w2d_split = 2: the headline workload (UNet 8 x 3x360x480, bench.py's seeds) with the forward, data-grad and weight-grad GEMMs of
    the 13 channel-heavy layers on fp16 split planes, against the REFERENCE-generated fixtures of the fp32 network: loss, dense logits (the
    frozen bound of the default path), gradient norms."""
    import json
    import os
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    G = os.path.join(os.path.dirname(__file__), "golden")
    d = dict(np.load(os.path.join(G, "unet_s0_8x360x480.npz")))
    meta = json.loads(str(d["meta"]))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(meta["data_seed"])
    x = torch.randn(8, 3, 360, 480, generator=g).to(dev); t = torch.randint(0, 12, (8, 360, 480), generator=g).to(dev)

    def run(split):
        torch.manual_seed(meta["seed"])
        net = A.UNet(3, 12).to(dev).train()
        runner_of(net).w2d_split = split
        out = net(x)
        loss = A.CrossEntropyLoss()(out, t)
        loss.backward()
        return net, out.detach(), float(loss)
    net, out, loss = run(2)
    assert abs(loss - float(d["loss"])) < 2e-5, (loss, float(d["loss"]))
    dd = np.load(os.path.join(G, "unet_s0_8x360x480_dense.npz"))
    ref = dd["logits_dense"]
    got = out[:, :, ::8, ::8].cpu().numpy()
    dv = np.abs(got - ref)
    mx, frac, rel = float(dv.max()), float((dv > 3e-4).mean()), float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    rel_g = []
    for i, (k, p) in enumerate(net.named_parameters()):
        if k.endswith("conv.0.bias"):
            continue
        assert torch.isfinite(p.grad).all(), k
        rel_g.append(abs(float(p.grad.double().norm()) - d["grad_l2"][i]) / d["grad_l2"][i])
    rel_g = np.array(rel_g)
    print(f"fp16-split mode, headline workload: loss {loss:.7f} (reference {float(d['loss']):.7f}); dense logits max |dev| {mx:.3e}, share beyond 3e-4 "
          f"{frac:.2e}, relative L2 {rel:.3e}; gradient norms: worst {rel_g.max():.2e}, median {np.median(rel_g):.2e}")
    tol = json.load(open(os.path.join(G, "drift.json")))["logits_tolerance"]["unet_s0_8x360x480"]["slice_abs"]
    assert mx <= tol and frac <= 5e-3 and rel <= 1.5e-4, (mx, frac, rel)
    assert rel_g.max() < 0.05 and np.median(rel_g) < 2e-3, (rel_g.max(), np.median(rel_g))
    _, out32, _ = run(0)
    dmode = float((out - out32).abs().max())
    assert 0.0 < dmode < 2 * tol, dmode
    # bitwise reproducible (the maximum passes use atomicMax: order-independent)
    _, outb, lossb = run(2)
    assert torch.equal(out, outb) and loss == lossb


def test_producer_passes_leave_the_exact_maximum():
    """cvk_bn_relu_apply_amax / cvk_bn_relu_apply_pool_amax / cvk_bn_bwd_dx_amax: the results are bitwise those of the plain entry points and the
    device word holds exactly the largest magnitude written (what cvk_absmax_f32 of the output returns)."""
    from pytorch_camvid_amd._lib import View
    lib, check = _lib()
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    N, H, W, C = 2, 23, 30, 64
    g = torch.Generator().manual_seed(9)
    y = (torch.randn(N, H, W, C, generator=g) * 3).to(dev)
    sc = (torch.rand(C, generator=g) + 0.5).to(dev); sh = (torch.randn(C, generator=g) * 0.3).to(dev)

    def view(t):
        return View(t.data_ptr(), H * W * C, W * C, C)
    out0 = torch.empty_like(y); out1 = torch.empty_like(y)
    a = _block(lib, dev)
    check(lib.cvk_bn_relu_apply(y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), view(out0), N, H, W, C, s), "apply")
    check(lib.cvk_bn_relu_apply_amax(y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), view(out1), N, H, W, C, a.data_ptr(), s), "apply amax")
    assert torch.equal(out0, out1) and _val(a) == out0.max().item()
    # pool variant: both words
    p0 = torch.empty(N, H // 2, W // 2, C, device=dev); p1 = torch.empty_like(p0)
    ao = _block(lib, dev); ap = _block(lib, dev)
    check(lib.cvk_bn_relu_apply_pool(y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), view(out0), p0.data_ptr(), None, N, H, W, C, s), "apply pool")
    check(lib.cvk_bn_relu_apply_pool_amax(y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), view(out1), p1.data_ptr(), None, N, H, W, C,
                                          ao.data_ptr(), ap.data_ptr(), s), "apply pool amax")
    assert torch.equal(out0, out1) and torch.equal(p0, p1)
    assert _val(ao) == out0.max().item() and _val(ap) == p0.max().item()
    # BatchNorm/ReLU backward: dy
    dO = torch.randn(N, H, W, C, generator=g).to(dev) * 1e-5
    mean = y.mean(dim=(0, 1, 2)).contiguous(); rstd = (1.0 / (y.var(dim=(0, 1, 2), unbiased=False) + 1e-5).sqrt()).contiguous()
    dgamma = torch.randn(C, generator=g).to(dev) * 1e-3; dbeta = torch.randn(C, generator=g).to(dev) * 1e-3
    PB = lib.cvk_bn_bwd_blocks(N * H * W)
    part0 = torch.zeros(PB * C, device=dev); part1 = torch.zeros(PB * C, device=dev)
    dy0 = torch.empty_like(y); dy1 = torch.empty_like(y)
    ad = _block(lib, dev)
    args = (y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr())
    check(lib.cvk_bn_bwd_dx(view(dO), *args, dy0.data_ptr(), C, part0.data_ptr(), N, H, W, C, 1, s), "bwd dx")
    check(lib.cvk_bn_bwd_dx_amax(view(dO), *args, dy1.data_ptr(), C, part1.data_ptr(), N, H, W, C, 1, ad.data_ptr(), s), "bwd dx amax")
    assert torch.equal(dy0, dy1) and torch.equal(part0, part1)
    assert _val(ad) == dy0.abs().max().item()


def test_public_switch_and_graph_replay_in_the_fp16_mode():
    """cvk.set_split_operands(net, 2) is the public switch (0 / 3 / 2, anything else is refused); in that mode the split kernels really run
    (engine profile names), and the step captured as ONE HIP graph — amax blocks zeroed, filled and read inside the graph — replays bit for
    bit what the eager step computes, over optimizer steps."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import engine
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev).train()
    ref = A.UNet(3, 12).to(dev).train()
    ref.load_state_dict(net.state_dict())
    with pytest.raises(ValueError):
        A.set_split_operands(net, 1)
    A.set_split_operands(net, 2); A.set_split_operands(ref, 2)
    lossf = A.CrossEntropyLoss()
    g = torch.Generator().manual_seed(3)

    def batch():
        return torch.randn(4, 3, 360, 480, generator=g).to(dev), torch.randint(0, 12, (4, 360, 480), generator=g).to(dev)

    def step(m, x, t):
        for p in m.parameters():
            p.grad = None
        loss = lossf(m(x), t)
        loss.backward()
        return loss.detach()
    x0, t0 = batch()
    engine.PROF = []
    try:
        step(ref, x0, t0)
        names = {r[0] for r in engine.PROF}
    finally:
        engine.PROF = None
    assert {"k_gemm_split2h", "k_gemm_tn_split2h", "k_conv3x3_wino4h", "k_conv3x3_wino4h<bnred>"} <= names, sorted(names)
    assert not any(n.startswith("k_w2d_gemm") or n == "k_conv3x3_wino4f" for n in names), sorted(names)
    ref.load_state_dict(net.state_dict())
    gs = A.GraphedStep(net, lossf, x0, t0)
    net.load_state_dict(ref.state_dict())
    oa, ob = torch.optim.SGD(net.parameters(), lr=0.05), torch.optim.SGD(ref.parameters(), lr=0.05)
    for it in range(2):
        x, t = batch()
        la, lb = gs.replay(x, t), step(ref, x, t)
        assert la.item() == lb.item(), it
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad), (it, k)
        oa.step(); ob.step()
