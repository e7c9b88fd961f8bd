"""Drop-in nn.Module surface of the reference's model zoo, executed by libcvk on MI355X.

Same constructor signatures, attribute tree, parameter registration order (= RNG consumption order, so
`torch.manual_seed(s); UNet(3, 12)` initialises bit-identically to the reference) and state_dict keys as
  models/unet.py:5-17  BasicConv2d      models/unet.py:19-32  UpSample2d      models/unet.py:35-156 UNet
  models/segnet.py:5-17 BasicConv       models/segnet.py:19-119 SegNet        utils.py:147-160      get_model
The torch.nn.Conv2d / BatchNorm2d children are kept ONLY as parameter/buffer containers (state_dict, .cuda(),
optimizers, checkpoints all work unchanged); their forward is never called.  forward() records a static plan for the
input geometry (engine.Plan) and replays it with hand-written HIP kernels; there is no eager/CPU fallback.
"""
import weakref

import torch
import torch.nn as nn

from . import engine
from .engine import BufView


def _channels_last_(conv):
    """Store OIHW weights physically as [Cout][3][3][Cin] (KRSC): the layout the implicit-GEMM kernels read, so
    no per-step re-layout of 138 MB of weights.  Logical shape and values are unchanged (state_dict compatible)."""
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)


class _Block(nn.Module):
    """Shared machinery of the two conv+BN+ReLU block flavours."""
    tag = "cbr"

    def conv_bn(self):
        raise NotImplementedError

    @property
    def in_channels(self):
        return self.conv_bn()[0].in_channels

    @property
    def out_channels(self):
        return self.conv_bn()[0].out_channels

    def block_params(self):
        c, b = self.conv_bn()
        return [c.weight, c.bias, b.weight, b.bias]

    def _emit(self, plan, src_buf, dst_view=None):
        return plan.conv_bn_relu(src_buf, self, dst_view)

    def forward(self, x):
        return _run(self, x)


class BasicConv2d(_Block):
    """Conv2d(3x3, pad 1, bias) + BatchNorm2d + ReLU; keys conv.0.*, conv.1.* (reference models/unet.py:5-17)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels),
                                  nn.ReLU(inplace=True))
        _channels_last_(self.conv[0])

    def conv_bn(self):
        return self.conv[0], self.conv[1]


class BasicConv(_Block):
    """Same block with SegNet's attribute names conv / bn / relu (reference models/segnet.py:5-17)."""

    def __init__(self, input_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(input_channels, out_channels, 3, padding=1)
        self.bn = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU()
        _channels_last_(self.conv)

    def conv_bn(self):
        return self.conv, self.bn


class UpSample2d(nn.Module):
    """Bilinear x2 (align_corners=True) followed by BasicConv2d (reference models/unet.py:19-32; the scale_factor
    argument is accepted and ignored exactly as there)."""

    def __init__(self, in_channels, out_channels, scale_factor=2.0):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = BasicConv2d(in_channels, out_channels)

    def _emit(self, plan, src_buf, dst_view=None):
        return self.conv._emit(plan, plan.upsample(src_buf), dst_view)

    def forward(self, x):
        return _run(self, x)


class _Stage(nn.Sequential):
    """nn.Sequential of blocks that can also emit itself into a plan."""

    def _emit(self, plan, src_buf, dst_view=None):
        n = len(self)
        v = None
        for i, blk in enumerate(self):
            v = blk._emit(plan, src_buf, dst_view if i == n - 1 else None)
            src_buf = v.buf
        return v

    def forward(self, x):
        return _run(self, x)


_UNET_WIDTHS = (64, 128, 256, 512, 1024)


class UNet(nn.Module):
    """reference models/unet.py:35-156.  Registration order: down1..down5, then per level upsample_k, up_k, then
    output, maxpool — the reference's order, so default init consumes the RNG identically."""

    def __init__(self, input_channels, class_num):
        super().__init__()
        cin = input_channels
        for k, w in enumerate(_UNET_WIDTHS, 1):
            self.add_module(f"down{k}", _Stage(BasicConv2d(cin, w), BasicConv2d(w, w)))
            cin = w
        for k in range(1, 5):
            wide, narrow = _UNET_WIDTHS[5 - k], _UNET_WIDTHS[4 - k]
            self.add_module(f"upsample{k}", UpSample2d(wide, narrow))
            self.add_module(f"up{k}", _Stage(BasicConv2d(wide, narrow), BasicConv2d(narrow, narrow)))
        self.output = BasicConv2d(_UNET_WIDTHS[0], class_num)
        self.maxpool = nn.MaxPool2d(2, 2)

    def _emit(self, plan, src_buf, dst_view=None):
        cats = []
        t = src_buf
        for k in range(1, 5):
            stage = getattr(self, f"down{k}")
            w = stage[-1].out_channels
            # the skip tensor is produced directly inside channels [w, 2w) of the level's concat buffer
            cat = plan.new_buf(2 * w, t.H, t.W, f"cat{k}")
            skip = stage._emit(plan, t, BufView(cat, w, w, 0, 0, t.H, t.W))
            cats.append((cat, skip))
            # fp32: the BN-apply pass that writes the pooled tensor also leaves the 1-byte arg-max codes, so the pool's backward reads
            # 0.25 B per input element instead of re-reading the 4-byte activations to recompute the arg-max
            t = plan.maxpool(skip, keep_code=not plan.bf16).dst
        t = self.down5._emit(plan, t).buf
        for k in range(1, 5):
            cat, skip = cats[4 - k]
            w = skip.C
            uh, uw = 2 * t.H, 2 * t.W
            dh, dw = cat.H - uh, cat.W - uw            # models/unet.py:117-123: pad [dw//2, dw-dw//2, dh//2, dh-dh//2]
            if dh < 0 or dw < 0:
                raise RuntimeError("upsampled branch larger than the skip tensor (negative padding is not supported)")
            win = BufView(cat, 0, w, dh // 2, dw // 2, uh, uw)
            if dh or dw:
                plan.zero_frame(win)
            getattr(self, f"upsample{k}")._emit(plan, t, win)
            t = getattr(self, f"up{k}")._emit(plan, cat).buf
        return self.output._emit(plan, t, dst_view)

    def forward(self, x):
        return _run(self, x)


_SEGNET = (("encoder1", (None, 64, 64)), ("encoder2", (64, 128, 128)), ("encoder3", (128, 256, 256, 256)),
           ("encoder4", (256, 512, 512, 512)), ("encoder5", (512, 512, 512, 512)),
           ("decoder5", (512, 512, 512, 512)), ("decoder4", (512, 512, 512, 256)), ("decoder3", (256, 256, 256, 128)),
           ("decoder2", (128, 128, 64)), ("decoder1", (64, 64, None)))


class SegNet(nn.Module):
    """reference models/segnet.py:19-119: 13-conv encoder with pooling indices, mirrored 13-conv decoder with unpooling."""

    def __init__(self, input_channels, class_num):
        super().__init__()
        for name, chain in _SEGNET:
            chain = [input_channels if c is None and i == 0 else (class_num if c is None else c) for i, c in enumerate(chain)]
            self.add_module(name, _Stage(*[BasicConv(a, b) for a, b in zip(chain[:-1], chain[1:])]))
        self.maxpool = nn.MaxPool2d(2, return_indices=True)
        self.unpool = nn.MaxUnpool2d(2)

    def _emit(self, plan, src_buf, dst_view=None):
        pools = []
        t = src_buf
        for k in range(1, 6):
            v = getattr(self, f"encoder{k}")._emit(plan, t)
            op = plan.maxpool(v, keep_code=True)
            pools.append(op)
            t = op.dst
        v = None
        for k in range(5, 0, -1):
            t = plan.unpool(t, pools[k - 1])
            v = getattr(self, f"decoder{k}")._emit(plan, t, dst_view if k == 1 else None)
            t = v.buf
        return v

    def forward(self, x):
        return _run(self, x)


def get_model(model_name, input_channels, class_num):
    """reference utils.py:147-160."""
    if model_name == "unet":
        return UNet(input_channels, class_num)
    if model_name == "segnet":
        return SegNet(input_channels, class_num)
    raise ValueError("network type does not supported")


# ------------------------------------------------------------------------------------------------- execution glue
def _holders_of(root):
    return [m for m in root.modules() if isinstance(m, _Block)]


_STATE = weakref.WeakKeyDictionary()   # module -> {"runner", "plans"}; kept out of the module so deepcopy/pickle stay plain


def _state_of(module):
    st = _STATE.get(module)
    if st is None:
        st = _STATE[module] = {"runner": engine.Runner(), "plans": {}}
        runner = st["runner"]

        def _loaded(_module, _incompatible):          # load_state_dict copies into the parameters: derived weights are stale
            runner.wepoch += 1
        module.register_load_state_dict_post_hook(_loaded)
    return st


def _run(root, x):
    if not isinstance(x, torch.Tensor) or x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got {tuple(x.shape) if isinstance(x, torch.Tensor) else type(x)}")
    if not x.is_cuda:
        raise RuntimeError("pytorch_camvid_amd runs on MI355X (HIP) tensors only; there is no CPU fallback. "
                           "Move the module and the input to 'cuda'.")
    if x.dtype != torch.float32:
        raise RuntimeError(f"expected float32 input (reference dtype), got {x.dtype}")
    if torch.jit.is_tracing():
        return _run_traced(root, x)
    state = _state_of(root)
    N, C, H, W = x.shape
    bf16 = bool(state["runner"].bf16)
    key = (N, C, H, W, bool(x.requires_grad and torch.is_grad_enabled()), bf16, bool(state["runner"].thin) if bf16 else None)
    plan = state["plans"].get(key)
    if plan is None:
        plan = engine.Plan(N, C, H, W, bf16=bf16)
        plan.input_needs_grad = key[4]
        plan.output = root._emit(plan, plan.input)
        plan.seal(thin=bool(state["runner"].thin))
        state["plans"][key] = plan
    params = []
    for h in plan.holders:
        params.extend(h.block_params())
    p0 = params[0]
    if p0.device != x.device or p0.dtype != torch.float32:
        raise RuntimeError(f"Input type ({x.dtype}, {x.device}) and weight type ({p0.dtype}, {p0.device}) should be the same")
    return engine.run_plan(state["runner"], plan, root.training, x, params)


def _run_traced(root, x):
    """torch.jit.trace support (reference train.py:97 -> utils.py:10-13 `writer.add_graph(net, tensor)` traces the
    module once at start-up).  The tracer cannot see kernels launched through the C ABI, so the network is executed
    with tracing suspended and its result enters the trace as ONE opaque constant: the trace (and its re-run check)
    succeeds, BatchNorm running statistics are updated exactly as the reference's traced passes update them, and the
    graph tensorboard draws has no inner layers."""
    import warnings
    warnings.warn("pytorch_camvid_amd: torch.jit.trace sees the HIP network as a single opaque constant "
                  "(writer.add_graph works but draws no inner layers)", stacklevel=3)
    ts = torch._C._get_tracing_state()
    torch._C._set_tracing_state(None)
    try:
        with torch.no_grad():
            out = _run(root, x.detach())
    finally:
        torch._C._set_tracing_state(ts)
    return out.contiguous(memory_format=torch.channels_last)


def set_conv_precision(module, precision):
    """"fp32" (default): exact-fp32 MFMA convolutions (Winograd where eligible), fp32 tensors everywhere.
    "bf16" (BASELINE.json configs[3]; UNet and SegNet, with or without a gradient for the input): activations and activation gradients are stored in HBM as
    bf16 NHWC, convolutions (forward, data-grad, weight-grad) multiply bf16 x bf16 on the matrix cores with fp32
    accumulation, BatchNorm statistics come from the fp32 accumulators; parameters, parameter gradients, BatchNorm
    parameters/buffers, logits and the loss stay fp32, so optimizers and checkpoints are unchanged.  Expect relative
    differences of ~1e-2 against the fp32 path (tests/golden/drift.json derives the stated tolerance)."""
    if precision not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    _state_of(module)["runner"].bf16 = precision == "bf16"
    return module


def set_split_operands(module, fmt):
    """OPT-IN (never the default; fp32 plans, training passes): run the matrix products of the fp32 convolutions on the 16-bit matrix pipe with
    SPLIT fp32 operands accumulated in fp32 (csrc/split_fmt.h):
      0  off — exact-fp32 MFMA everywhere (the default);
      3  three bf16 terms per operand, six cross-products: the 2-D Winograd GEMMs of the channel-heavy layers;
      2  two fp16 terms per operand scaled by an exact power of two (from the tensors' largest magnitudes, left in device memory by the
         passes that write them), three cross-products: those GEMMs AND the fused F(4,3) convolutions of the 64/128-channel levels.
    Tensors, parameters, statistics and checkpoints stay fp32; measured against the reference fixtures both forms are at least as close as
    the default path (tests/test_gpu_split2h.py, tests/test_gpu_split3.py, tests/test_gpu_protocol.py).  Same as CVK_W2D_SPLIT in the
    environment before the first forward pass."""
    if fmt not in (0, 2, 3):
        raise ValueError("fmt must be 0 (off), 3 (bf16 x 3) or 2 (fp16 x 2)")
    _state_of(module)["runner"].w2d_split = fmt
    return module


def runner_of(module):
    """The engine.Runner of a module (created on first use); ddp.DataParallel hooks gradient sync into it."""
    return _state_of(module)["runner"]
