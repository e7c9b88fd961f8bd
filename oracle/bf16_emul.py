"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU emulation of WHERE the bf16-storage mode (set_conv_precision(net, "bf16"); csrc/conv_bf16s.hip, csrc/elem_bf16.hip)
rounds, on top of the stock-torch rebuild of the reference graphs (oracle/torch_ref.py; reference models/unet.py:5-32,
94-156 and models/segnet.py:5-119).  Arithmetic is fp32 as on the device (bf16 x bf16 products are exact in fp32, accumulation is fp32); every
tensor the device keeps in HBM as bf16 is rounded to bf16 (round-to-nearest-even) at the point it is stored:
  forward : the imported input, conv weights (a bf16 copy of the fp32 masters), the pre-BN conv output y, the BN+ReLU
            activation a, the bilinear-upsampled tensor; pooled tensors are maxima of rounded values (exact);
  backward: the gradient entering y (the device's dy) and the gradient entering a / the upsampled tensor (its dX).
The logits, the loss, the loss gradient, BatchNorm statistics, parameters and parameter gradients stay fp32.
Used to DERIVE the tolerance of the bf16 parity tests (tests/golden/make_drift.py -> tests/golden/drift.json): the
distance between this emulation and the fp32 run of the same graph is what bf16 storage costs by construction.
"""
import torch
import torch.nn.functional as F


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """Stored as bf16 in both directions: value rounded in forward, incoming gradient rounded in backward."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundFwd(torch.autograd.Function):
    """bf16 copy of an fp32 master tensor: rounded value, gradient passes unrounded to the master."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


def _cbr(conv, bn, x, last=False):
    y = _RoundBoth.apply(F.conv2d(x, _RoundFwd.apply(conv.weight), conv.bias, padding=1))
    z = F.relu(bn(y))
    return z if last else _RoundBoth.apply(z)


def _stage(seq, x, last=False):
    n = len(seq)
    for i, blk in enumerate(seq):
        x = _cbr(blk.conv[0], blk.conv[1], x, last and i == n - 1)
    return x


def unet_forward(net, x):
    """oracle.torch_ref.RefUNet.forward with the device's rounding points (reference models/unet.py:94-156)."""
    x = _r(x)
    skips = []
    for k in range(1, 5):
        x = _stage(getattr(net, f"down{k}"), x)
        skips.append(x)
        x = F.max_pool2d(x, 2, 2)
    x = _stage(net.down5, x)
    for k in range(1, 5):
        skip = skips[4 - k]
        up = getattr(net, f"upsample{k}")
        u = _RoundBoth.apply(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))
        u = _cbr(up.conv.conv[0], up.conv.conv[1], u)
        dh = skip.size(2) - u.size(2); dw = skip.size(3) - u.size(3)
        u = F.pad(u, [dw // 2, dw - dw // 2, dh // 2, dh - dh // 2])
        x = _stage(getattr(net, f"up{k}"), torch.cat([u, skip], dim=1))
    return _cbr(net.output.conv[0], net.output.conv[1], x, last=True)


def segnet_forward(net, x):
    """oracle.torch_ref.RefSegNet.forward with the device's rounding points (reference models/segnet.py:84-119).  Pooling and
    unpooling move bf16 values without arithmetic (exact); the arg-max is taken on the stored (rounded) activations, as the
    device recomputes it from them.  The gradient entering an unpooled tensor is a conv data-grad, stored as bf16: rounded."""
    x = _r(x)
    marks = []
    for k in range(1, 6):
        for blk in getattr(net, f"encoder{k}"):
            x = _cbr(blk.conv, blk.bn, x)
        shape = x.shape
        x, idx = F.max_pool2d(x, 2, return_indices=True)
        marks.append((idx, shape))
    for k in range(5, 0, -1):
        idx, shape = marks[k - 1]
        x = _RoundBoth.apply(F.max_unpool2d(x, idx, 2, output_size=shape))
        seq = getattr(net, f"decoder{k}")
        for i, blk in enumerate(seq):
            x = _cbr(blk.conv, blk.bn, x, last=(k == 1 and i == len(seq) - 1))
    return x


def basic_conv(block, x, last=False):
    """One conv+BN+ReLU block (oracle.torch_ref._CBR) in the emulation; x is rounded on entry like an imported input."""
    return _cbr(block.conv[0], block.conv[1], _r(x), last)


def fwd_bwd_step(net, x, t, model="unet"):
    for p in net.parameters():
        p.grad = None
    out = unet_forward(net, x) if model == "unet" else segnet_forward(net, x)
    loss = F.cross_entropy(out, t)
    loss.backward()
    return loss, out
