"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Stock-`torch.nn` CPU rebuild of the two reference networks, table-driven.  The reference's hot path
*is* stock torch.nn (no native code, every FLOP is ATen), so the faithful CPU
comparator is the same operators wired the same way.  This file restates the wiring from
  models/unet.py:37-92 (module tree / construction order), :94-156 (forward),
  models/segnet.py:19-80 (module tree), :82-119 (forward),
  utils.py:147-160 (get_model), train.py:100-134 (training step),
with the same state_dict key names and the same RNG consumption order, so that
`torch.manual_seed(s); build_unet(3, 12)` is bit-identical to the reference's
`torch.manual_seed(s); UNet(3, 12)` (pinned by tests/test_oracle_golden.py against logits that
tests/golden/make_golden.py produced by importing the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# (stage name, [(cin, cout), ...]) in CONSTRUCTION ORDER == RNG consumption order
_UNET_ENC = [("down1", None, 64), ("down2", 64, 128), ("down3", 128, 256), ("down4", 256, 512), ("down5", 512, 1024)]
_UNET_DEC = [(1, 1024, 512), (2, 512, 256), (3, 256, 128), (4, 128, 64)]
_SEGNET = [
    ("encoder1", [(None, 64), (64, 64)]),
    ("encoder2", [(64, 128), (128, 128)]),
    ("encoder3", [(128, 256), (256, 256), (256, 256)]),
    ("encoder4", [(256, 512), (512, 512), (512, 512)]),
    ("encoder5", [(512, 512), (512, 512), (512, 512)]),
    ("decoder5", [(512, 512), (512, 512), (512, 512)]),
    ("decoder4", [(512, 512), (512, 512), (512, 256)]),
    ("decoder3", [(256, 256), (256, 256), (256, 128)]),
    ("decoder2", [(128, 128), (128, 64)]),
    ("decoder1", [(64, 64), (64, None)]),
]


class _CBR(nn.Module):
    """conv3x3(pad 1) + BN + ReLU with the UNet key naming `conv.{0,1}` (models/unet.py:5-17)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class _CBRSeg(nn.Module):
    """Same block with the SegNet key naming `conv` / `bn` (models/segnet.py:5-17)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _CBR(cin, cout)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))


class RefUNet(nn.Module):
    def __init__(self, input_channels, class_num):
        super().__init__()
        for name, cin, cout in _UNET_ENC:
            cin = input_channels if cin is None else cin
            self.add_module(name, nn.Sequential(_CBR(cin, cout), _CBR(cout, cout)))
        for k, cin, cout in _UNET_DEC:
            self.add_module(f"upsample{k}", _Up(cin, cout))
            self.add_module(f"up{k}", nn.Sequential(_CBR(cin, cout), _CBR(cout, cout)))
        self.output = _CBR(64, class_num)

    def forward(self, x):
        skips = []
        for k in range(1, 5):
            x = getattr(self, f"down{k}")(x)
            skips.append(x)
            x = F.max_pool2d(x, 2, 2)
        x = self.down5(x)
        for k in range(1, 5):
            skip = skips[4 - k]
            u = getattr(self, f"upsample{k}")(x)
            dh = skip.size(2) - u.size(2); dw = skip.size(3) - u.size(3)
            u = F.pad(u, [dw // 2, dw - dw // 2, dh // 2, dh - dh // 2])
            x = getattr(self, f"up{k}")(torch.cat([u, skip], dim=1))
        return self.output(x)


class RefSegNet(nn.Module):
    def __init__(self, input_channels, class_num):
        super().__init__()
        for name, blocks in _SEGNET:
            mods = []
            for cin, cout in blocks:
                cin = input_channels if cin is None else cin
                cout = class_num if cout is None else cout
                mods.append(_CBRSeg(cin, cout))
            self.add_module(name, nn.Sequential(*mods))

    def forward(self, x):
        marks = []
        for k in range(1, 6):
            x = getattr(self, f"encoder{k}")(x)
            shape = x.shape
            x, idx = F.max_pool2d(x, 2, return_indices=True)
            marks.append((idx, shape))
        for k in range(5, 0, -1):
            idx, shape = marks[k - 1]
            x = F.max_unpool2d(x, idx, 2, output_size=shape)
            x = getattr(self, f"decoder{k}")(x)
        return x


def build(model_name, input_channels, class_num):
    """utils.py:147-160 get_model semantics."""
    if model_name == "unet":
        return RefUNet(input_channels, class_num)
    if model_name == "segnet":
        return RefSegNet(input_channels, class_num)
    raise ValueError("network type does not supported")


def synthetic_batch(n, h, w, seed=1234, classes=12, channels=3):
    """SURVEY §8d synthetic inputs: randn images, randint labels, CPU generator seeded `seed`."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, channels, h, w, generator=g)
    t = torch.randint(0, classes, (n, h, w), generator=g)
    return x, t


def fwd_bwd_step(net, x, t):
    """The timed region of SURVEY §8d on the CPU comparator: zero_grad -> net(x) -> CE -> backward."""
    for p in net.parameters():
        p.grad = None
    loss = F.cross_entropy(net(x), t)
    loss.backward()
    return loss
