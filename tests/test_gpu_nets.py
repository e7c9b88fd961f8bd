"""Whole-network parity on the GPU: UNet / SegNet against reference goldens (weights by recipe), eval mode,
optimizer trajectory, full-size (BASELINE.json configs) properties."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
NETS = ["unet_s0_2x48x64", "unet_s1_1x45x60", "unet_s2_2x36x52", "segnet_s0_2x64x96", "segnet_s3_2x45x60", "segnet_s4_2x96x128"]


def dev():
    return torch.device("cuda:0")


def batch(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g)
    t = torch.randint(0, 12, (n, h, w), generator=g)
    return x.to(dev()), t.to(dev())


def _acts(net, x):
    """Every activation buffer of one training-mode forward pass of `net`, by buffer id (engine.RunState.act), + the plan."""
    from pytorch_camvid_amd import engine
    from pytorch_camvid_amd.modules import _state_of
    with torch.no_grad():
        net(x)                                            # records the plan for this geometry
    state = _state_of(net)
    plan = [p for k, p in state["plans"].items() if k[:4] == tuple(x.shape)][0]
    params = []
    for h in plan.holders:
        params.extend(h.block_params())
    _, st = state["runner"].forward(plan, x, params, True, True)
    torch.cuda.synchronize()
    return plan, st


def _assert_deviation_is_an_argmax_tie(tag, meta, conv_setup, x):
    """ADVICE r3 / VERDICT r3 2d: a forced kernel mode that moves a SegNet golden by more than the tolerance must be EXPLAINED, not
    skipped.  Run the default mode and the forced mode on identical weights and input and walk the activation buffers in execution
    order.  Claim checked: up to the first MaxUnpool2d (models/segnet.py:104-116) whose output differs, the two modes agree to
    rounding; at that unpool the outputs differ because max-pool arg-max indices differ (models/segnet.py:79), and in EVERY window
    whose index differs the two largest inputs lie closer together than the modes' own rounding-level difference on that tensor —
    a tie that either summation order may break.  A defect in the forced kernels (wrong tile walk, wrong fused pool code, wrong
    BN sums) would show as a difference that is NOT born at such a tie and fails here."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import engine
    nets = []
    for forced in (False, True):
        torch.manual_seed(meta["seed"])
        net = A.get_model(meta["kind"], 3, 12).to(dev()).train()
        if forced:
            conv_setup(net)
        nets.append(net)
    (plan, sa), (plan_b, sb) = _acts(nets[0], x), _acts(nets[1], x)
    assert len(plan.ops) == len(plan_b.ops)
    first = None
    for op in plan.ops:
        if isinstance(op, engine.MaxPool) and op.keep_code:
            ca, cb = sa.saved[op.idx], sb.saved[op.idx]                     # 1-byte arg-max codes of every window
            if not torch.equal(ca, cb):
                first = op
                break
        elif isinstance(op, engine.ConvBnRelu):
            a, b = sa.act[op.dst.buf.id].float(), sb.act[op.dst.buf.id].float()
            scale = max(a.abs().max().item(), 1e-6)
            # before the first flip the modes agree to accumulated rounding (BatchNorm over as few as 6-24 samples amplifies it)
            assert (a - b).abs().max().item() <= 2e-3 * scale, (tag, "conv block output differs before any pool index differs", op.idx,
                                                               (a - b).abs().max().item(), scale)
    assert first is not None, (tag, "logits differ although every max-pool index agrees")
    pool = first
    pre_a, pre_b = sa.act[pool.src.buf.id].float(), sb.act[pool.src.buf.id].float()          # the max pool's input, both modes
    rounding = (pre_a - pre_b).abs().max().item()
    d = pool.dst
    N, H, W, C = pre_a.shape[0], d.H * 2, d.W * 2, pool.src.C
    win = pre_a[:, :H, :W, :C].reshape(N, d.H, 2, d.W, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(N, d.H, d.W, C, 4)
    top2 = win.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1])                                                        # [N, H/2, W/2, C]
    moved = (sa.saved[pool.idx] != sb.saved[pool.idx]).view(N, d.H, d.W, d.ld)[..., :C]
    nmoved = int(moved.sum().item())
    assert 0 < nmoved <= max(4, int(2e-3 * moved.numel())), (tag, "windows with a different arg-max", nmoved, moved.numel())
    worst = gap[moved].max().item()
    print(f"{tag}: {nmoved} of {moved.numel()} windows of the max pool at op {pool.idx} flipped; largest top-2 gap there {worst:.3e}, "
          f"mode-to-mode rounding on that tensor {rounding:.3e}")
    assert worst <= 4.0 * rounding + 1e-7, (tag, "an arg-max moved although its window was not a tie", worst, rounding)


@pytest.mark.parametrize("tag,conv", [(t, c) for t in NETS for c in ("default", "f43_always")] +
                         [(t, c) for t in NETS if t.startswith("unet") for c in ("w2d_always", "w2d4_always")])
def test_net_forward_loss_grads_golden(tag, conv):
    """conv = "default": the engine's per-layer choice (direct / F(2,3) / F(4,3) by grid size; at these small goldens
    mostly F(2,3)).  "f43_always": every eligible layer through the F(4,3) kernels, logits tolerance 1e-3 instead of 5e-4:
    F(4,3) rounds ~2.5x coarser than F(2,3) (6e-7 vs 2.5e-7 relative rms per layer, csrc/wino4.hip) and these tiny
    geometries amplify a per-layer relative perturbation ~800x (BatchNorm over 6-12 samples at the bottleneck; measured
    with oracle/torch_ref in fp64 + injected noise).  tests/chaos_probe.py over 8 data seeds: direct 3e-4, F(2,3) 2.5e-4,
    F(4,3) 5e-4 max logits deviation.  SegNet's pool/unpool pairs are discontinuous in the arg-max: F(4,3) rounding flips one on
    exactly one of the three small fixtures, and WHICH one depends on the kernel's summation order — round 2's per-index F(4,3)
    kernels flipped segnet_s0 (31 % of the logits move, the other two within 4.8e-4), round 3's fused kernel (csrc/wino4f.hip)
    passes s0 and s3 within 4.8e-4 and flips segnet_s4 (44 % move; the per-index kernels pass it).  Round 4: no fixture is skipped any
    more.  A SegNet case in a forced mode that exceeds the logits tolerance must pass _assert_deviation_is_an_argmax_tie (the
    deviation is born at a max-pool window whose two largest inputs are closer than the modes' rounding difference, and nowhere
    before); the loss / gradient / running-statistics checks below then do not apply to it (everything downstream of the unpool
    differs).  The reference's own 1e-6 input perturbation does NOT flip these three small fixtures (tests/golden/*_perturbed.npz,
    max 4.3e-4), so they stay pinned at 5e-4 in the default mode.  The raw kernels are pinned against fp64 at the bottleneck
    geometries in tests/test_gpu_wino4f.py.
    "w2d_always" / "w2d4_always": every eligible layer (>= 32 input, >= 64 output channels) through the 2-D F(6x6,3x3) (the default
    tile for UNet since round 3; about twice the rounding of F(4x4), held to the same tolerance) / F(4x4,3x3) kernels, which the
    engine itself only uses from 256 tiles up; they round 3.5x coarser again (2.8e-6 vs 8e-7 relative L2 per layer,
    tests/test_drift_cpu.py::test_winograd2d_rounding), hence 3.5 x the F(4,3) logits tolerance.  UNet only: in SegNet the coarser
    rounding flips max-pool arg-max indices even on the 2x96x128 fixture (unpooling is discontinuous in them); SegNet's layers
    reach the 2-D kernels only at full size, where the per-kernel raw-ABI tests (test_gpu_large.py) and the engine's default
    choice in the SegNet bench cover them."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    d = dict(np.load(os.path.join(G, tag + ".npz")))
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = A.get_model(meta["kind"], 3, 12)
    assert [k for k, _ in net.named_parameters()] == list(d["param_names"])
    net = net.to(dev()).train()
    if conv == "f43_always":
        runner_of(net).wino4 = "always"
        runner_of(net).wgradp = "always"      # ... and every eligible weight-grad through the transform-domain planes (csrc/wgradp.hip)
    if conv in ("w2d_always", "w2d4_always"):
        runner_of(net).wino2d = "always"
        runner_of(net).w2tile_cfg = 4 if conv == "w2d4_always" else 6
    n, _, h, w = meta["shape"]
    x, t = batch(n, h, w, meta["data_seed"])
    out = net(x)
    assert tuple(out.shape) == (n, 12, h, w)
    # forward tolerance (fp32, 23-26 conv+BN layers, BN over as few as 6-12 samples at the bottleneck of these
    # small goldens): 5e-4 absolute on logits in [0, ~5]; the reference itself moves ~5e-5 between fp32 and fp64
    atol = {"default": 5e-4, "f43_always": 1e-3, "w2d_always": 3.5e-3, "w2d4_always": 3.5e-3}[conv]
    got = out.detach().cpu().numpy()
    if meta["kind"] == "segnet" and conv != "default" and not np.allclose(got, d["logits"], rtol=1e-3, atol=atol):
        def setup(m):
            runner_of(m).wino4 = "always"
            runner_of(m).wgradp = "always"
        _assert_deviation_is_an_argmax_tie(tag, meta, setup, x)
        return
    np.testing.assert_allclose(got, d["logits"], rtol=1e-3, atol=atol)
    if meta["kind"] == "segnet" and os.path.exists(os.path.join(G, tag + "_perturbed.npz")):
        # the reference's own twin (1e-6 relative input noise): the device may not be further from the golden than 2 x that + 2e-4
        tw = np.load(os.path.join(G, tag + "_perturbed.npz"))
        ref_move = float(np.abs(tw["logits"] - d["logits"]).max())
        assert float(np.abs(got - d["logits"]).max()) <= 2.0 * ref_move + (2e-4 if conv == "default" else 6e-4), (tag, conv, ref_move)
    loss = A.CrossEntropyLoss()(out, t)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-5
    # gradients: the reference's own fp32 run deviates from fp64 by up to ~18 % per tensor at these tiny bottlenecks
    # (tests/test_oracle_golden.py); per-operator gradient parity is pinned tightly in test_gpu_blocks.py
    rel = []
    for i, (k, p) in enumerate(net.named_parameters()):
        assert p.grad is not None and p.grad.shape == p.shape, k
        if k.endswith("conv.0.bias") or k.endswith(".conv.bias"):
            assert p.grad.abs().max().item() < 1e-4, k
            continue
        gl2 = float(p.grad.double().norm())
        rel.append(abs(gl2 - d["grad_l2"][i]) / d["grad_l2"][i])
    rel = np.array(rel)
    assert rel.max() < 0.25 and np.mean(rel < 1e-2) >= 0.8, rel
    sd = net.state_dict()
    for k in sd:
        if "running_mean" in k or "running_var" in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), d["bn." + k], rtol=2e-4, atol=2e-5, err_msg=k)
    net.eval()
    with torch.no_grad():
        oe = net(x)
    assert abs(oe.double().sum().item() - float(d["logits_eval_sum"])) < 1e-4 * abs(float(d["logits_eval_sum"])) + 1e-2
    np.testing.assert_allclose(oe[0, :, ::7, ::5].cpu().numpy(), d["logits_eval_slice"], rtol=1e-3, atol=3e-4)
    agree = (A.argmax_channels(oe).cpu().numpy() == d["argmax_eval"]).mean()
    assert agree > 0.995, agree


@pytest.mark.parametrize("tag", ["unet_s0_2x48x64", "segnet_s0_2x64x96"])
def test_adamw_trajectory_golden(tag):
    """train.py:100-134 semantics with our network: AdamW + OneCycleLR, same seeds -> same loss curve."""
    import pytorch_camvid_amd as A
    d = dict(np.load(os.path.join(G, tag + ".npz")))
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = A.get_model(meta["kind"], 3, 12).to(dev()).train()
    n, _, h, w = meta["shape"]
    x, t = batch(n, h, w, meta["data_seed"])
    opt = torch.optim.AdamW(net.parameters(), lr=meta["lr"], weight_decay=0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=meta["lr"], total_steps=meta["total_steps"])
    lossf = A.CrossEntropyLoss()
    losses = []
    for _ in range(meta["steps"]):
        opt.zero_grad()
        l = lossf(net(x), t)
        l.backward()
        opt.step(); sched.step()
        losses.append(l.item())
    # Adam divides by sqrt(v): rounding-level gradient differences become lr-sized parameter differences, so the
    # trajectory is only as reproducible as the reference graph itself is.  The per-step tolerance is DERIVED:
    # tests/golden/make_drift.py measures fp32-vs-fp64 and fp32-vs-(1e-6 input noise, four seeds) loss drift of the
    # reference graph for exactly these runs; tolerance[i] = max(2e-5, 4 x the largest drift) (tests/golden/drift.json,
    # re-measured by tests/test_drift_cpu.py).
    tol = json.load(open(os.path.join(G, "drift.json")))["trajectory_tolerance"][tag]
    for i, (a, b) in enumerate(zip(losses, d["traj_losses"])):
        assert abs(a - b) < tol[i], (i, a, b)


# Regression sentinels for the full-size logits: measured device deviations from the reference on the DENSE fixtures
# (every 8th pixel: 30x the points of the slice), per 2-D Winograd tile.  The hard bound stays the derived tolerance of drift.json
# (4 x the reference graph's own drift, frozen by tests/test_drift_cpu.py); these sit ~1.1-1.3x above what the kernels measure today and BELOW the
# frozen bound (max-abs: 6.0e-4 against 6.6e-4 since round 5), so a coarser kernel trips here before it reaches the bound.  (max |dev|, share of points beyond 3e-4, relative L2)
DENSE_SENTINEL = {6: (6.0e-4, 4e-3, 1.3e-4), 4: (4.3e-4, 2e-4, 8.5e-5)}       # measured r4: 6 -> 5.3e-4 / 1.7e-3 / 9.6e-5, 4 -> 3.3e-4 / 2.3e-5 / 6.3e-5


def _dense_check(out, dense_tag, tile):
    dd = np.load(os.path.join(G, dense_tag + ".npz"))
    ref = dd["logits_dense"]
    got = out[:, :, ::8, ::8].detach().cpu().numpy()
    dv = np.abs(got - ref)
    mx, frac, rel = float(dv.max()), float((dv > 3e-4).mean()), float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    npix = out.shape[0] * out.shape[2] * out.shape[3]
    o64 = out.detach().double()
    dmean = float(np.abs(o64.sum(dim=(0, 2, 3)).cpu().numpy() - dd["class_sum"]).max() / npix)
    dsq = float(np.abs((o64 ** 2).sum(dim=(0, 2, 3)).cpu().numpy() - dd["class_sq_sum"]).max() / npix)
    edges = torch.tensor(dd["hist_edges"], device=out.device, dtype=torch.float32)
    tv = 0.0
    for c in range(out.shape[1]):
        h = torch.histogram(out[:, c].detach().float().cpu().flatten(), bins=edges.cpu())[0].numpy()
        tv = max(tv, float(np.abs(h - dd["class_hist"][c]).sum() / (2.0 * npix)))
    print(f"{dense_tag} tile {tile}: max |dev| {mx:.3e}, share beyond 3e-4 {frac:.2e}, relative L2 {rel:.3e} over {ref.size} points; "
          f"per-class mean shift {dmean:.2e}, mean-square shift {dsq:.2e}, histogram distance {tv:.2e}")
    smx, sfrac, srel = DENSE_SENTINEL[tile]
    assert mx <= smx and frac <= sfrac and rel <= srel, (dense_tag, tile, mx, frac, rel)
    # a systematic shift of a class (which max-abs over samples cannot see) and its distribution
    assert dmean <= 2e-5 and dsq <= 1e-4 and tv <= 2e-3, (dmean, dsq, tv)


def test_segnet_forced_2d_mode_deviation_is_an_argmax_tie():
    """The coarsest forced mode (every eligible layer through 2-D F(6x6,3x3), which the engine never picks at this size) on the
    SegNet fixture it is known to disturb: either the logits still meet the forced-mode tolerance, or the deviation must be
    explained as a max-pool tie by _assert_deviation_is_an_argmax_tie (keeps that analysis exercised)."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    tag = "segnet_s4_2x96x128"
    d = dict(np.load(os.path.join(G, tag + ".npz")))
    meta = json.loads(str(d["meta"]))

    def setup(m):
        runner_of(m).wino2d = "always"
        runner_of(m).w2tile_cfg = 6
    torch.manual_seed(meta["seed"])
    net = A.get_model("segnet", 3, 12).to(dev()).train()
    setup(net)
    n, _, h, w = meta["shape"]
    x, t = batch(n, h, w, meta["data_seed"])
    with torch.no_grad():
        got = net(x).cpu().numpy()
    if np.allclose(got, d["logits"], rtol=1e-3, atol=3.5e-3):
        return
    _assert_deviation_is_an_argmax_tie(tag, meta, setup, x)


@pytest.mark.parametrize("tile", [6, 4])
def test_unet_fullsize_batch8_dense_logits(tile):
    """Headline workload, train-mode logits against the dense reference fixture, for the default 2-D tile F(6x6,3x3) and for
    F(4x4,3x3) (CVK_W2D_TILE=4 / runner.w2tile_cfg = 4: the finer-rounding mode; bound = the pre-round-3 3e-4 on the slice)."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    d = dict(np.load(os.path.join(G, "unet_s0_8x360x480.npz")))
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = A.UNet(3, 12).to(dev()).train()
    runner_of(net).w2tile_cfg = tile
    x, t = batch(8, 360, 480, meta["data_seed"])
    with torch.no_grad():
        out = net(x)
    _dense_check(out, "unet_s0_8x360x480_dense", tile)
    sh, sw = meta["slice"]
    sl = np.abs(out[:, :, ::sh, ::sw].cpu().numpy() - d["logits_slice"]).max()
    if tile == 4:
        assert sl <= 3e-4, sl          # the fixed slice bound of rounds 1-2, still met by the 4x4 tile


def test_unet_fullsize_batch2_golden():
    """BASELINE.json configs[0] geometry (2x3x360x480): loss, logits checksum/slice, grad norms vs the reference."""
    import pytorch_camvid_amd as A
    d = dict(np.load(os.path.join(G, "unet_s0_2x360x480.npz")))
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    x, t = batch(2, 360, 480, 1234)
    out = net(x)
    loss = A.CrossEntropyLoss()(out, t)
    loss.backward()
    assert abs(loss.item() - float(d["traj_losses"][0])) < 2e-5
    assert abs(out.double().sum().item() - float(d["logits_sum"])) < 2e-5 * float(d["logits_abs_sum"])
    # slice tolerance DERIVED (tests/golden/make_drift.py logits): 4 x the drift of the reference graph's own logits between fp32 and
    # fp64 / 1e-6 input noise (BatchNorm's division by the channel deviation turns rounding into 1e-4 on the logits): 6.7e-4
    atol = json.load(open(os.path.join(G, "drift.json")))["logits_tolerance"]["unet_s0_2x360x480"]["slice_abs"]
    np.testing.assert_allclose(out[:, :, ::40, ::48].detach().cpu().numpy(), d["logits_slice"], rtol=1e-3, atol=atol)
    _dense_check(out, "unet_s0_2x360x480_dense", 6)
    names = list(d["param_names"])
    rel = []
    for i, (k, p) in enumerate(net.named_parameters()):
        if k.endswith("conv.0.bias"):
            continue
        rel.append(abs(float(p.grad.double().norm()) - d["grad_l2"][i]) / d["grad_l2"][i])
    rel = np.array(rel)
    assert rel.max() < 0.05 and np.median(rel) < 2e-3, (rel.max(), np.median(rel))


def test_unet_fullsize_batch8_golden():
    """BASELINE.json configs[1], the headline workload (UNet 8x3x360x480, bench.py's seeds): loss, logits checksums and
    slice, per-parameter gradient norms and slices against the reference-generated fixture unet_s0_8x360x480.npz."""
    import pytorch_camvid_amd as A
    d = dict(np.load(os.path.join(G, "unet_s0_8x360x480.npz")))
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = A.UNet(3, 12).to(dev()).train()
    x, t = batch(8, 360, 480, meta["data_seed"])
    out = net(x)
    loss = A.CrossEntropyLoss()(out, t)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-5
    assert abs(out.double().sum().item() - float(d["logits_sum"])) < 2e-5 * float(d["logits_abs_sum"])
    assert abs((out.double() ** 2).sum().item() - float(d["logits_sq_sum"])) < 1e-4 * float(d["logits_sq_sum"])
    sh, sw = meta["slice"]
    # slice tolerance DERIVED from the reference graph's own logits drift at this workload (tests/golden/make_drift.py logits): 6.6e-4
    atol = json.load(open(os.path.join(G, "drift.json")))["logits_tolerance"]["unet_s0_8x360x480"]["slice_abs"]
    sl = out[:, :, ::sh, ::sw].detach().cpu().numpy()
    np.testing.assert_allclose(sl, d["logits_slice"], rtol=1e-3, atol=atol)
    # ... and in L2 no further from the reference than 4 x the distance between two evaluations of the reference itself
    drift = json.load(open(os.path.join(G, "drift.json")))["logits"]["unet_s0_8x360x480"]
    assert np.linalg.norm(sl - d["logits_slice"]) / np.linalg.norm(d["logits_slice"]) < 4.0 * max(drift["fp64_rel_l2"], drift["noise_rel_l2"])
    names = list(d["param_names"])
    assert [k for k, _ in net.named_parameters()] == names
    rel, sl = [], []
    for i, (k, p) in enumerate(net.named_parameters()):
        if k.endswith("conv.0.bias"):
            continue
        rel.append(abs(float(p.grad.double().norm()) - d["grad_l2"][i]) / d["grad_l2"][i])
        if p.dim() == 4:
            gs = p.grad.permute(0, 1, 2, 3).contiguous().flatten()[:64].cpu().numpy()      # logical OIHW order, as the fixture
            sl.append(np.abs(gs - d["gs." + k]).max() / (d["grad_absmax"][i] + 1e-30))
    rel = np.array(rel); sl = np.array(sl)
    # 8x more BatchNorm samples than the batch-2 golden: the whole-net gradients are correspondingly better conditioned
    assert rel.max() < 0.03 and np.median(rel) < 1e-3, (rel.max(), np.median(rel))
    # element-wise slices: ReLU-mask flips between two fp32 runs move single elements by ~1 % of the tensor's largest entry
    assert np.median(sl) < 2e-2 and sl.max() < 0.1, (np.median(sl), sl.max())
    bm = np.array([float(v.double().norm()) for k, v in net.state_dict().items() if k.endswith("running_mean")])
    bv = np.array([float(v.double().norm()) for k, v in net.state_dict().items() if k.endswith("running_var")])
    np.testing.assert_allclose(bm, d["bn_mean_l2"], rtol=1e-4)
    np.testing.assert_allclose(bv, d["bn_var_l2"], rtol=1e-4)


@pytest.mark.parametrize("tile", [6, 4])
def test_unet_fullsize_batch8_dense_gradients(tile):
    """VERDICT r5 #3 — the gradients of the headline workload (BASELINE.json configs[1]; loss.backward() of train.py:130-131), ELEMENT-WISE
    against the imported reference (tests/golden/unet_s0_8x360x480_grads.npz, make_golden.py b8grads): the full weight gradient of six conv
    layers of the full- / half-resolution double-conv blocks and the head, every 97th element of every other conv weight gradient, every
    BatchNorm gamma / beta gradient — for the default 2-D tile F(6x6,3x3) and for F(4x4,3x3).  Tolerances per tensor are DERIVED
    (tests/golden/make_drift.py grads -> drift.json grads_tolerance): 4 x the larger of the reference graph's own fp32-vs-fp64 and
    fp32-vs-1e-6-input-noise drift of that tensor (relative L2, and largest element difference over the tensor's largest entry)."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    d = dict(np.load(os.path.join(G, "unet_s0_8x360x480_grads.npz")))
    meta = json.loads(str(d["meta"]))
    dj = json.load(open(os.path.join(G, "drift.json")))
    tol, drift = dj["grads_tolerance"]["unet_s0_8x360x480"], dj["grads"]["unet_s0_8x360x480"]
    torch.manual_seed(meta["seed"])
    net = A.UNet(3, 12).to(dev()).train()
    runner_of(net).w2tile_cfg = tile
    x, t = batch(8, 360, 480, meta["data_seed"])
    loss = A.CrossEntropyLoss()(net(x), t)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-5
    used, rows = [], []
    for k, p in net.named_parameters():
        if k.endswith("conv.0.bias"):
            continue
        got = p.grad.detach().contiguous().flatten()                 # logical OIHW order, as the fixture
        if p.dim() == 4 and k not in meta["dense"]:
            got = got[::meta["stride"]]
        got = got.double().cpu().numpy()
        ref = d["g." + k].astype(np.float64)
        assert got.shape == ref.shape, k
        rl2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        mx = float(np.abs(got - ref).max() / np.abs(ref).max())
        rows.append((k, rl2, mx, tol[k]["rel_l2"], tol[k]["max_rel"]))
        used.append(max(rl2 / tol[k]["rel_l2"], mx / tol[k]["max_rel"]))
        # also relative to the reference's own drift (1.0 = as far from the reference as the reference is from itself)
    worst = int(np.argmax(used))
    dense = [r for r in rows if r[0] in meta["dense"]]
    print(f"dense gradients, tile {tile}: {len(rows)} tensors ({sum(d['g.' + r[0]].size for r in rows)} elements), share of the derived tolerance "
          f"used: max {max(used):.2f} ({rows[worst][0]}), median {float(np.median(used)):.2f}; six full tensors: "
          + ", ".join(f"{r[0].split('.conv.0')[0]} relL2 {r[1]:.1e}/{r[3]:.1e}" for r in dense))
    for (k, rl2, mx, t1, t2) in rows:
        assert rl2 <= t1 and mx <= t2, (k, rl2, t1, mx, t2, "reference drift", drift[k])


def test_unet_batch8_properties():
    """BASELINE.json configs[1] (8x3x360x480): size-independent properties — per-sample independence of eval-mode
    forward (batch of 8 == eight batches of 1), determinism (bitwise equal reruns), finite grads for all 92 tensors."""
    import pytorch_camvid_amd as A
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev())
    x, t = batch(8, 360, 480, 1234)
    net.train()
    lossf = A.CrossEntropyLoss()
    l1 = lossf(net(x), t); l1.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    l2 = lossf(net(x), t); l2.backward()
    assert l1.item() == l2.item()
    for a, p in zip(g1, net.parameters()):
        assert torch.isfinite(p.grad).all()
        assert torch.equal(a, p.grad)                       # no atomics anywhere: bitwise reproducible
    net.eval()
    with torch.no_grad():
        full = net(x)
        for i in (0, 5):
            one = net(x[i:i + 1])
            assert torch.allclose(full[i:i + 1], one, rtol=1e-4, atol=1e-5)


def test_data_parallel_rccl_plumbing_single_rank():
    """RCCL path on one GPU: world_size-1 'nccl' process group with the collectives forced on.  Gradients must be
    bitwise those of the plain run (AVG over one rank), buckets must tile the flat buffer, and the all-reduces must
    be issued while backward is still running (before finish)."""
    import torch.distributed as dist
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import ddp
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:             # a free port, not a fixed one: two suites on one host must not collide
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev())
    try:
        torch.manual_seed(0)
        net = A.UNet(3, 12).to(dev()).train()
        x, t = batch(2, 96, 128, 5)
        lossf = A.CrossEntropyLoss()
        lossf(net(x), t).backward()
        ref = [p.grad.clone() for p in net.parameters()]
        for p in net.parameters():
            p.grad = None
        # undo the running-stat update of the first pass so both passes see identical state
        dp = ddp.DataParallel(net, bucket_mb=8.0, always_issue=True)
        lossf(dp(x), t).backward()
        torch.cuda.synchronize()
        for a, p in zip(ref, net.parameters()):
            assert torch.equal(a, p.grad)
        launched = dp.sync.launched
        assert len(launched) >= 4 and launched[0][0] == 0
        for (a0, a1), (b0, b1) in zip(launched[:-1], launched[1:]):
            assert a1 == b0
        total = sum((p.numel() + 3) // 4 * 4 for p in net.parameters())
        assert launched[-1][1] == total
    finally:
        dist.destroy_process_group()


def test_flat_adamw_matches_torch_adamw():
    """§8f row 2: the fused flat AdamW step against torch.optim.AdamW on identical gradients, with OneCycleLR driving both."""
    import pytorch_camvid_amd as A
    torch.manual_seed(0)
    n1 = A.UNet(3, 12).to(dev()).train()
    n2 = A.UNet(3, 12).to(dev()).train()
    n2.load_state_dict(n1.state_dict())
    x, t = batch(2, 48, 64, 9)
    o1 = torch.optim.AdamW(n1.parameters(), lr=5e-4, weight_decay=1e-2)
    o2 = A.FlatAdamW(n2, lr=5e-4, weight_decay=1e-2)
    s1 = torch.optim.lr_scheduler.OneCycleLR(o1, max_lr=5e-4, total_steps=20)
    s2 = torch.optim.lr_scheduler.OneCycleLR(o2, max_lr=5e-4, total_steps=20)
    lossf = A.CrossEntropyLoss()
    for it in range(3):
        for net, opt, sch in ((n1, o1, s1), (n2, o2, s2)):
            opt.zero_grad()
            lossf(net(x), t).backward()
        # identical weights -> identical gradients (bitwise deterministic kernels); compare the updates
        for (k, a), (_, b) in zip(n1.named_parameters(), n2.named_parameters()):
            if it == 0:
                assert torch.equal(a.grad, b.grad), k
        o1.step(); s1.step(); o2.step(); s2.step()
        for (k, a), (_, b) in zip(n1.named_parameters(), n2.named_parameters()):
            if k.endswith("conv.0.bias"):
                continue    # noise-level gradients: Adam's sign(g) makes these chaotic in any implementation
            # step 1 is checked element-wise to 1e-7 below.  Afterwards 1e-7 weight differences feed back through the
            # gradients, and Adam's m/sqrt(v) turns sign flips of near-zero gradient elements into lr-sized (5e-4)
            # element moves in ANY two runs, so later steps only bound the element difference by a few learning rates
            assert (a - b).abs().max().item() < 4 * 5e-4, (it, k, (a - b).abs().max().item())
        if it == 0:
            for (k, a), (_, b) in zip(n1.named_parameters(), n2.named_parameters()):
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (k, (a - b).abs().max().item())
    assert n2.down1[0].conv[0].weight.is_contiguous(memory_format=torch.channels_last)
    assert len(n2.state_dict()) == 161


def test_preprocess_and_evaluate():
    import pytorch_camvid_amd as A
    from oracle import np_ops as O
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (2, 36, 52, 3), generator=g, dtype=torch.uint8)
    x = A.preprocess_uint8(img.to(dev()))
    mean = torch.tensor(A.functional.CAMVID_MEAN).view(1, 3, 1, 1); std = torch.tensor(A.functional.CAMVID_STD).view(1, 3, 1, 1)
    want = (img.permute(0, 3, 1, 2).float() / 255 - mean) / std            # transforms.py:485-538 arithmetic
    assert tuple(x.shape) == (2, 3, 36, 52)
    assert torch.allclose(x.cpu(), want, rtol=1e-6, atol=1e-6)
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    net(x)                                                                   # one training pass fills the running stats
    masks = torch.randint(0, 12, (2, 36, 52), generator=g).to(dev())
    acc, iou, miou = A.evaluate(net, [(x, masks), (x, masks)])
    assert net.training
    net.eval()
    with torch.no_grad():
        pred = net(x).argmax(dim=1).cpu().numpy()
    _, _, miou_o = O.mean_iou([pred[0], pred[1]] * 2, [masks.cpu().numpy()[0], masks.cpu().numpy()[1]] * 2)
    assert abs(miou - miou_o) < 1e-9


def test_segnet_fullsize_batch8_golden():
    """BASELINE.json configs[4] (SegNet 8x3x360x480, /root/reference/models/segnet.py:82-119, bench.py's seeds) against the
    reference-generated fixture segnet_s0_8x360x480.npz — at this size the network runs the fused F(4,3) and the 2-D
    F(4x4,3x3) kernels the engine picks for the full-size workload.  SegNet's five pool/unpool pairs are discontinuous in the
    max-pool arg-max, so single logits and single gradient entries are NOT reproducible even by the reference itself: the
    fixture segnet_s0_8x360x480_perturbed.npz is the same reference run under a 1e-6 relative input perturbation (11 % of
    its sampled logits move by > 1e-3, the largest by 0.65; loss by 1.9e-5).  Every tolerance below is 4 x the distance
    between the two reference runs (floors: the UNet batch-8 tolerances), so the device has to sit as close to the reference
    as the reference sits to itself."""
    import pytorch_camvid_amd as A
    d = dict(np.load(os.path.join(G, "segnet_s0_8x360x480.npz")))
    q = dict(np.load(os.path.join(G, "segnet_s0_8x360x480_perturbed.npz")))
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = A.SegNet(3, 12).to(dev()).train()
    x, t = batch(8, 360, 480, meta["data_seed"])
    out = net(x)
    loss = A.CrossEntropyLoss()(out, t)
    loss.backward()

    def tol(key, floor, scale=1.0):
        return max(floor, 4.0 * abs(float(d[key]) - float(q[key])) / scale)
    assert abs(loss.item() - float(d["loss"])) < tol("loss", 2e-5), (loss.item(), float(d["loss"]))
    asum = float(d["logits_abs_sum"])
    assert abs(out.double().sum().item() - float(d["logits_sum"])) < asum * tol("logits_sum", 2e-5, asum)
    sq = float(d["logits_sq_sum"])
    assert abs((out.double() ** 2).sum().item() - sq) < sq * tol("logits_sq_sum", 1e-4, sq)
    sh, sw = meta["slice"]
    got = out[:, :, ::sh, ::sw].detach().cpu().numpy()
    dev_ = np.abs(got - d["logits_slice"]); own = np.abs(q["logits_slice"] - d["logits_slice"])
    # sampled logits: no further from the reference than its perturbed twin is, in distribution (arg-max flips move a tenth of them)
    assert np.mean(dev_ > 1e-3) <= max(0.02, 2.0 * np.mean(own > 1e-3)), (np.mean(dev_ > 1e-3), np.mean(own > 1e-3))
    assert dev_.max() <= max(3e-4, 2.0 * own.max()) and np.median(dev_) <= max(3e-4, 4.0 * np.median(own))
    names = list(d["param_names"])
    assert [k for k, _ in net.named_parameters()] == names
    rel, own = [], []
    for i, (k, p) in enumerate(net.named_parameters()):
        if k.endswith(".conv.bias"):      # conv bias under train-mode BN: mathematically zero, numerically noise
            continue
        ref = float(d["grad_l2"][i])
        rel.append(abs(float(p.grad.double().norm()) - ref) / ref)
        own.append(abs(float(q["grad_l2"][i]) - ref) / ref)
    rel, own = np.array(rel), np.array(own)
    # Which tensor an arg-max flip hits is chance (the reference's perturbed twin: median 0.19 %, 90th percentile 1.0 %, worst
    # tensor 4.3 %), so the comparison is between the two DISTRIBUTIONS of per-tensor gradient-norm deviations, not tensor by tensor
    assert np.median(rel) <= max(1e-3, 4.0 * np.median(own)), (np.median(rel), np.median(own))
    assert np.percentile(rel, 90) <= max(1e-2, 4.0 * np.percentile(own, 90)), (np.percentile(rel, 90), np.percentile(own, 90))
    assert rel.max() <= max(0.03, 2.0 * own.max()), (rel.max(), own.max())
    bm = np.array([float(v.double().norm()) for k, v in net.state_dict().items() if k.endswith("running_mean")])
    bv = np.array([float(v.double().norm()) for k, v in net.state_dict().items() if k.endswith("running_var")])
    np.testing.assert_allclose(bm, d["bn_mean_l2"], rtol=max(1e-4, 4.0 * float(np.max(np.abs(q["bn_mean_l2"] - d["bn_mean_l2"]) / d["bn_mean_l2"]))))
    np.testing.assert_allclose(bv, d["bn_var_l2"], rtol=max(1e-4, 4.0 * float(np.max(np.abs(q["bn_var_l2"] - d["bn_var_l2"]) / d["bn_var_l2"]))))


def test_large_geometries():
    """BASELINE.json configs[3] geometry (4x3x720x960, fp32 here) and configs[4] (SegNet 8x3x360x480): run, finite,
    deterministic, per-sample independence in eval mode."""
    import pytorch_camvid_amd as A
    lossf = A.CrossEntropyLoss()
    for kind, shape in (("unet", (4, 720, 960)), ("segnet", (8, 360, 480))):
        torch.manual_seed(0)
        net = A.get_model(kind, 3, 12).to(dev()).train()
        x, t = batch(shape[0], shape[1], shape[2], 11)
        l1 = lossf(net(x), t); l1.backward()
        g1 = [p.grad.clone() for p in net.parameters()]
        for p in net.parameters():
            p.grad = None
        l2 = lossf(net(x), t); l2.backward()
        assert torch.isfinite(l1) and l1.item() == l2.item()
        assert all(torch.equal(a, p.grad) and torch.isfinite(p.grad).all() for a, p in zip(g1, net.parameters()))
        net.eval()
        with torch.no_grad():
            full = net(x[:2]); one = net(x[1:2])
        assert torch.allclose(full[1:2], one, rtol=1e-4, atol=2e-5)
        del net, x, t, g1
        torch.cuda.empty_cache()


def test_eval_report_and_predict_equivalents():
    """reference eval.py:44-80 (mIoU, precision, recall, mean loss) and predict.py:35-57 (normalise -> eval forward ->
    argmax -> nearest resize) restated on the device, against numpy restatements of legacy/metrics.py:20-71 and of
    cv2's INTER_NEAREST index rule."""
    import pytorch_camvid_amd as A
    torch.manual_seed(0)
    net = A.get_model("unet", 3, 12).to(dev()).eval()
    batches = [batch(2, 48, 64, 40 + i) for i in range(3)]
    rep = A.evaluate_report(net, batches, num_classes=12, ignore_index=11)
    cm = np.zeros((12, 12)); losses = []
    with torch.no_grad():
        for x, t in batches:
            lg = net(x)
            losses.append(torch.nn.functional.cross_entropy(lg.float(), t).item())
            p = lg.argmax(1).cpu().numpy().ravel(); g = t.cpu().numpy().ravel()
            keep = g != 11                                         # utils.intersect_and_union masks ignore_index pixels
            np.add.at(cm, (g[keep], p[keep]), 1)
    valid = [c for c in range(12) if c != 11]
    diag = np.diag(cm)
    iou = diag / (cm.sum(1) + cm.sum(0) - diag + 1e-15)
    assert abs(rep["miou"] - iou[valid].mean()) < 1e-9
    assert abs(rep["precision"] - (diag / (cm.sum(0) + 1e-15))[valid].mean()) < 1e-9
    assert abs(rep["recall"] - (diag / (cm.sum(1) + 1e-15))[valid].mean()) < 1e-9
    assert abs(rep["loss"] - np.mean(losses)) < 1e-5
    # predict: uint8 BGR frame -> class map; equals argmax of the eval forward on the normalised frame
    g = torch.Generator().manual_seed(3)
    frame = torch.randint(0, 256, (48, 64, 3), generator=g, dtype=torch.uint8)
    cls = A.predict(net, frame.numpy())
    with torch.no_grad():
        want = net(A.preprocess_uint8(frame.to(dev()).unsqueeze(0))).argmax(1)[0]
    assert cls.dtype == torch.int64 and torch.equal(cls, want)
    big = A.predict(net, frame, out_size=(100, 130))
    yi = np.minimum(np.floor(np.arange(100) * (48 / 100)).astype(int), 47)
    xi = np.minimum(np.floor(np.arange(130) * (64 / 130)).astype(int), 63)
    assert np.array_equal(big.cpu().numpy(), want.cpu().numpy()[yi][:, xi])
    with pytest.raises(ValueError):
        A.predict(net, torch.zeros(48, 64, 3))                     # not uint8


def test_device_prefetcher_matches_direct_upload():
    """pinned, one-batch-ahead uint8 upload + device-side normalisation == the direct path, batch by batch."""
    import pytorch_camvid_amd as A
    g = torch.Generator().manual_seed(9)
    host = [(torch.randint(0, 256, (2, 24, 32, 3), generator=g, dtype=torch.uint8).numpy(),
             torch.randint(0, 12, (2, 24, 32), generator=g)) for _ in range(5)]
    got = list(A.DevicePrefetcher(host))
    assert len(got) == 5
    for (x, m), (f, t) in zip(got, host):
        want = A.preprocess_uint8(torch.from_numpy(f).to(dev()))
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (2, 3, 24, 32)
        assert torch.equal(x, want) and torch.equal(m.cpu(), t)
    with pytest.raises(ValueError):
        list(A.DevicePrefetcher([(torch.zeros(2, 24, 32, 3), torch.zeros(2, 24, 32, dtype=torch.long))]))


@pytest.mark.parametrize("kind", ["unet", "segnet"])
def test_eval_mode_backward_runs_the_round6_paths(kind):
    """Fine-tuning with frozen BatchNorm (net.eval(), gradients on): the fused forward launch has no statistics to compute but still leaves the
    weight-grad's V planes, the BatchNorm-backward pass runs without batch statistics and writes four E planes, the plane GEMM reads E0 / E5 from dy.
    Against the stock-torch network (oracle/torch_ref.py) in eval mode with the same running statistics, at a geometry where those paths are
    active (2 x 96 x 128: 6144 tile rows at full resolution).  Loss to 2e-5; every gradient within 4 x the reference graph's own fp32-vs-fp64 distance
    (derived in the test: ReLU masks and pool arg-maxes are discontinuous in eval mode too)."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import engine
    from oracle import torch_ref as R
    torch.manual_seed(4)
    ref = R.build(kind, 3, 12).train()
    torch.manual_seed(4)
    net = A.get_model(kind, 3, 12).to(dev()).train()
    x, t = batch(2, 96, 128, 77)
    with torch.no_grad():
        ref(x.cpu())                                           # one training-mode pass: running statistics away from (0, 1)
    net.load_state_dict(ref.state_dict())
    net.eval(); ref.eval()
    engine.PROF = []
    try:
        out = net(x)
        loss = A.CrossEntropyLoss()(out, t)
        loss.backward()
        torch.cuda.synchronize()
        names = {e[0] for e in engine.PROF}
    finally:
        engine.PROF = None
    assert "k_conv3x3_wino4f<vplanes>" in names and "k_bn_bwd<dx+E4p>" in names and "k_wgradp_gemm" in names, sorted(names)
    lr = torch.nn.functional.cross_entropy(ref(x.cpu()), t.cpu())
    lr.backward()
    assert abs(loss.item() - lr.item()) < 2e-5
    # the tolerance is DERIVED in place: ReLU masks and pool arg-maxes are discontinuous in eval mode too, so the reference graph's own fp32 run is
    # compared with its fp64 run (same weights, same statistics) and the device may be 4 x as far from fp64 as that (floor 3e-4)
    import copy
    r64 = copy.deepcopy(ref).double()
    for p_ in r64.parameters():
        p_.grad = None
    torch.nn.functional.cross_entropy(r64(x.cpu().double()), t.cpu()).backward()
    worst = (0.0, "", 0.0)
    for (k, p), q, q64 in zip(net.named_parameters(), ref.parameters(), r64.parameters()):
        g, w32, w = p.grad.detach().cpu().double(), q.grad.double(), q64.grad
        if w.norm() == 0:
            assert g.abs().max() < 1e-8, k
            continue
        drift = float((w32 - w).norm() / w.norm())
        rel = float((g - w).norm() / w.norm())
        tol = max(4.0 * drift, 3e-4)
        if rel / tol > worst[0]:
            worst = (rel / tol, k, rel)
        assert rel <= tol, (k, rel, drift)
    print(f"eval-mode backward, {kind}: worst share of the derived tolerance {worst[0]:.2f} ({worst[1]}: {worst[2]:.2e})")
