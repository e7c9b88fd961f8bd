"""Host-side executor logic that needs no GPU: plan recording (op order, zero-copy concat views, pad windows),
flat gradient layout, kernel-name mirror of the C dispatch."""
import torch

import pytorch_camvid_amd as A
from pytorch_camvid_amd import engine


def build_plan(net, n, c, h, w):
    plan = engine.Plan(n, c, h, w)
    plan.output = net._emit(plan, plan.input)
    plan.seal()
    return plan


def test_unet_plan_structure_360x480():
    net = A.UNet(3, 12)
    p = build_plan(net, 8, 3, 360, 480)
    kinds = [type(o).__name__ for o in p.ops]
    assert kinds.count("ConvBnRelu") == 23 and kinds.count("MaxPool") == 4 and kinds.count("Upsample") == 4
    assert kinds.count("ZeroFrame") == 1                      # only level 1 pads at 360x480 (45 -> 22 -> 44 -> pad 45)
    assert len(p.holders) == 23
    # execution order of the conv blocks == construction order of the reference (models/unet.py:37-92)
    names = [k for k, _ in net.named_modules() if isinstance(_, A.BasicConv2d)]
    order = [names[[m for _, m in net.named_modules() if isinstance(m, A.BasicConv2d)].index(h)] for h in p.holders]
    assert order[:4] == ["down1.0", "down1.1", "down2.0", "down2.1"] and order[10] == "upsample1.conv" and order[-1] == "output"
    # zero-copy concat: the skip conv writes channels [C,2C) of the cat buffer, the upsample conv channels [0,C) through a window
    convs = [o for o in p.ops if isinstance(o, engine.ConvBnRelu)]
    d41 = convs[7]                                            # down4.1
    assert d41.dst.buf.C == 1024 and (d41.dst.c0, d41.dst.C, d41.dst.H, d41.dst.W) == (512, 512, 45, 60)
    up1 = convs[10]                                           # upsample1.conv: 44x60 window at the top of the 45x60 buffer
    assert up1.dst.buf is d41.dst.buf and (up1.dst.c0, up1.dst.y0, up1.dst.x0, up1.dst.H, up1.dst.W) == (0, 0, 0, 44, 60)
    zf = [o for o in p.ops if isinstance(o, engine.ZeroFrame)][0]
    assert zf.view.buf is d41.dst.buf and p.ops.index(zf) < p.ops.index(up1)
    assert convs[11].src is d41.dst.buf                       # up1.0 reads the whole concat buffer (1024 channels)
    out = p.output
    assert (out.buf.N, out.buf.H, out.buf.W, out.C) == (8, 360, 480, 12)
    assert p.input.ld == 4 and convs[0].src_needs_grad is False and all(c.src_needs_grad for c in convs[1:])


def test_segnet_plan_structure_and_errors():
    net = A.SegNet(3, 12)
    p = build_plan(net, 2, 3, 45, 60)
    kinds = [type(o).__name__ for o in p.ops]
    assert kinds.count("ConvBnRelu") == 26 and kinds.count("MaxPool") == 5 and kinds.count("Unpool") == 5
    pools = [o for o in p.ops if isinstance(o, engine.MaxPool)]
    assert all(o.keep_code for o in pools) and [(o.dst.H, o.dst.W) for o in pools] == [(22, 30), (11, 15), (5, 7), (2, 3), (1, 1)]
    unpools = [o for o in p.ops if isinstance(o, engine.Unpool)]
    assert [(o.dst.H, o.dst.W) for o in unpools] == [(2, 3), (5, 7), (11, 15), (22, 30), (45, 60)]   # output_size = encoder shapes
    import pytest
    with pytest.raises(RuntimeError):
        build_plan(A.UNet(3, 12), 1, 3, 15, 15)               # too small for four 2x2 pools (reference: pool RuntimeError)
    with pytest.raises(RuntimeError):
        build_plan(A.UNet(3, 12), 1, 5, 32, 32)               # channel mismatch, like F.conv2d


def test_flat_gradient_layout_is_reverse_execution_order():
    net = A.UNet(3, 12)
    p = build_plan(net, 1, 3, 32, 32)
    params = [t for h in p.holders for t in h.block_params()]
    offs, total = engine.layout_grads(params)
    assert offs[4 * 22] == 0                                  # the output block (last executed) sits at the front
    ends = [offs[i] + (params[i].numel() + 3) // 4 * 4 for i in range(len(params))]
    # blocks are contiguous and ordered last-executed-first; every view is 16-byte aligned
    for slot in range(22, 0, -1):
        assert ends[4 * slot + 3] == offs[4 * (slot - 1)]
    assert all(o % 4 == 0 for o in offs) and total == ends[3]
    assert total >= sum(x.numel() for x in params) == 34533924


def test_kernel_name_mirror():
    assert engine.conv_kernel_name("wino", 128) == "k_conv3x3_wino<128, 128, 2, 2>"
    assert engine.conv_kernel_name("wino", 64) == "k_conv3x3_wino<128, 64, 2, 2>"
    assert engine.conv_kernel_name("wino", 12) == "k_conv3x3_wino<128, 32, 4, 1>"
    assert engine.conv_kernel_name("fwd", 64, 4) == "k_conv3x3_igemm<128, 64, 2, 2, true, false>"
    assert engine.conv_kernel_name("dgrad", 64, 12) == "k_conv3x3_igemm<128, 64, 2, 2, false, false>"
    assert engine.conv_kernel_name("wgrad", 64, 4) == "k_conv3x3_wgrad<64, 64, 2, 2>"
    assert engine.conv_kernel_name("wgrad", 12, 64) == "k_wgrad_smallco"          # the 12-class head: 16x16x4-MFMA kernel
    assert engine.conv_kernel_name("wgrad", 12, 128) == "k_conv3x3_wgrad<32, 256, 1, 4>"
    assert engine.conv_kernel_name("wino4", 128) == "k_conv3x3_wino4<128, 128, 2, 2>"
    assert engine.conv_kernel_name("wino4", 64) == "k_conv3x3_wino4<128, 64, 2, 2>"


def test_winograd_variant_choice_per_layer():
    """engine.wino4_pays: F(4,3) wherever the layer is more than one wave of 512 resident workgroups, F(2,3) for the
    12-column head and for tiny geometries (the goldens) unless a test forces "always"."""
    pays = engine.wino4_pays
    assert pays(8, 360, 480, 64, 64) and pays(8, 360, 480, 128, 64) and pays(8, 180, 240, 128, 128)
    assert pays(8, 90, 120, 256, 256) and pays(8, 45, 60, 512, 512) and pays(8, 44, 60, 1024, 512)
    assert pays(8, 22, 30, 512, 1024) and pays(8, 22, 30, 1024, 1024)              # bottleneck: F(4,3) with a 3-way K split
    assert not pays(8, 360, 480, 64, 12)                                           # the logits layer
    assert not pays(2, 6, 8, 512, 512) and not pays(1, 45, 60, 64, 64)             # golden-sized layers
    assert pays(50, 360, 480, 64, 64)                                              # large batches


def test_winograd2d_choice_per_layer():
    """engine.wino2d_pays / wgrad2d_pays: the 2-D F(4x4,3x3) path takes the channel-heavy layers of the batch-8 step
    (measured per layer, tools/bench_conv.py) and leaves the 64/128-channel levels and golden-sized geometries to F(4,3)."""
    f, w = engine.wino2d_pays, engine.wgrad2d_pays
    for shape in ((8, 90, 120, 256, 256), (8, 45, 60, 256, 512), (8, 45, 60, 512, 512), (8, 22, 30, 512, 1024),
                  (8, 22, 30, 1024, 1024), (8, 44, 60, 1024, 512), (8, 45, 60, 1024, 512), (8, 90, 120, 512, 256)):
        assert f(*shape) and w(*shape), shape
    assert f(8, 180, 240, 256, 128) and f(8, 180, 240, 128, 256)                  # forward / data-grad only: many tiles
    assert not w(8, 180, 240, 256, 128)                                            # weight-grad there only with the forward's V
    for shape in ((8, 360, 480, 64, 64), (8, 360, 480, 128, 64), (8, 180, 240, 128, 128), (8, 180, 240, 64, 128),
                  (8, 90, 120, 128, 256), (8, 360, 480, 64, 12)):
        assert not f(*shape) and not w(*shape), shape
    assert not f(2, 6, 8, 512, 512) and not w(2, 6, 8, 512, 512)                  # golden-sized layers: fewer than 256 tiles
    # 6x6 tiles (UNet's default since round 3, tools/bench_w6.py): the same layer set (taking the 128-channel layers too gained
    # nothing on the step and doubled the logits deviation: engine.wino2d_pays)
    for shape in ((8, 90, 120, 256, 256), (8, 22, 30, 1024, 1024), (8, 180, 240, 256, 128), (8, 180, 240, 128, 128), (8, 90, 120, 128, 256),
                  (8, 360, 480, 64, 64), (8, 180, 240, 64, 128), (2, 6, 8, 512, 512)):
        assert engine.wino2d_pays(*shape, tile=6) == f(*shape), shape
    assert engine.wino2d_ok(256, 256, 256) and not engine.wino2d_ok(48, 256, 256) and not engine.wino2d_ok(256, 12, 12)


def test_winograd2d_tile_choice():
    """engine.layer_tile / Runner.tile_for: 6x6 output tiles (F(6x6,3x3)) unless the layer's tile count fills the GEMM's 128-row tiles
    badly (the 22x30 bottleneck at batch 8: 160 tiles) or the network unpools (SegNet: 4x4 in the forward pass, 6x6 for data-grads)."""
    class R:
        w2tile, w2tile_dgrad = 6, 6
    assert engine.layer_tile(R, 8, 22, 30) == 4 and engine.layer_tile(R, 8, 45, 60) == 6 and engine.layer_tile(R, 8, 90, 120) == 6
    assert engine.layer_tile(R, 8, 180, 240) == 6 and engine.layer_tile(R, 2, 45, 60) == 4
    R.w2tile = 4
    assert engine.layer_tile(R, 8, 90, 120) == 4 and engine.layer_tile(R, 8, 90, 120, dgrad=True) == 6
    R.w2tile_dgrad = 4
    assert engine.layer_tile(R, 8, 90, 120, dgrad=True) == 4


def test_which_layers_emit_v_planes_from_the_forward_pass():
    """Round 6: the fused F(4,3) forward launch of exactly the eight 64/128-input-channel layers of the headline step (the double-conv blocks of
    reference models/unet.py:40-47, 81-89 and their neighbours) leaves the weight-grad's V planes behind; the 2-D layers (V kept from their own
    input transform), the thin stem / head and the split-operand modes do not; no gradient pass, no planes."""
    from pytorch_camvid_amd.modules import runner_of

    class St:
        need_grad = True
    net = A.UNet(3, 12)
    p = build_plan(net, 8, 3, 360, 480)
    R = runner_of(net)
    R.w2tile, R.w2tile_dgrad = R.tile_for(p)
    convs = [o for o in p.ops if isinstance(o, engine.ConvBnRelu)]
    names = [k for k, m in net.named_modules() if isinstance(m, A.BasicConv2d)]
    order = [names[[m for _, m in net.named_modules() if isinstance(m, A.BasicConv2d)].index(h)] for h in p.holders]
    got = [n for n, c in zip(order, convs) if c._want_planes(R, St)]
    assert got == ["down1.1", "down2.0", "down2.1", "down3.0", "up3.1", "upsample4.conv", "up4.0", "up4.1"], got
    assert not any(c._want_planes(R, St) and c._wgrad2d(R) for c in convs)
    St.need_grad = False
    assert not any(c._want_planes(R, St) for c in convs)
    St.need_grad = True
    R.vplanes = False
    assert not any(c._want_planes(R, St) for c in convs)
    R.vplanes = True
    R.w2d_split = 2
    assert not any(c._want_planes(R, St) for c in convs)
    R.w2d_split = 0
    # tiny geometries keep the in-kernel transform (the plane GEMM needs >= 4096 tile rows)
    p2 = build_plan(A.UNet(3, 12), 2, 3, 48, 64)
    assert not any(c._want_planes(R, St) for c in p2.ops if isinstance(c, engine.ConvBnRelu))
    assert engine.vplanes_pays(8, 360, 480, 128, 64) and not engine.vplanes_pays(8, 360, 480, 96, 64) and not engine.vplanes_pays(64, 720, 960, 128, 64)
